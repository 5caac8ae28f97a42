"""Host mirror of the reference's generic online wrapper: forward(x (B, n_in)) -> (B, 368) in raw physical units.

  * `MLP`      online_testing/baseline_models/MLP_v2rh/training/mlp.py:25-67 (same constructor arguments; holds the
               weights as CPU tensors under the reference's state_dict keys `linears.{i}.0.{weight,bias}`,
               `final_linear.{weight,bias}`)
  * `NewModel` online_testing/model_postprocessing/v4_nn_wrapper.ipynb cell 5 (same constructor arguments and the same
               hard-wired column ranges of the v2_rh / v4 flat input vector)

All arithmetic runs in the HIP library (`csa_online_*`, csrc/online.hip); there is no CPU fallback.  Unlike the
reference wrapper the caller's `x` is not normalised in place."""
import ctypes

import numpy as np
import torch

from . import _lib
from .emulator import _check, _ptr


class MLP:
    """Weights container with the reference MLP's constructor; dropout is inference-inactive and therefore ignored."""

    def __init__(self, in_dims, out_dims, hidden_dims, layers, dropout=0., output_prune=False, strato_lev_out=15):
        if isinstance(hidden_dims, (list, tuple)):
            assert len(hidden_dims) == layers, "Length of hidden_dims should be equal to layers"
            hidden_dims = list(hidden_dims)
        else:
            hidden_dims = [hidden_dims] * layers
        self.in_dims, self.out_dims, self.hidden_dims, self.layers = in_dims, out_dims, hidden_dims, layers
        self.output_prune, self.strato_lev_out = output_prune, strato_lev_out
        dims = [in_dims] + hidden_dims + [out_dims]
        g = torch.Generator().manual_seed(0)
        self._state = {}
        for i in range(layers + 1):
            k = 1.0 / np.sqrt(dims[i])                  # nn.Linear's default init range
            w = (torch.rand(dims[i + 1], dims[i], generator=g) * 2 - 1) * k
            b = (torch.rand(dims[i + 1], generator=g) * 2 - 1) * k
            name = f"linears.{i}.0" if i < layers else "final_linear"
            self._state[name + ".weight"], self._state[name + ".bias"] = w, b

    def state_dict(self):
        return dict(self._state)

    def load_state_dict(self, sd):
        for k, v in self._state.items():
            t = torch.as_tensor(sd[k], dtype=torch.float32).detach().cpu().contiguous()
            if t.shape != v.shape:
                raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {tuple(v.shape)}")
            self._state[k] = t

    def weights(self):
        names = [f"linears.{i}.0" for i in range(self.layers)] + ["final_linear"]
        return [self._state[n + ".weight"].numpy() for n in names], [self._state[n + ".bias"].numpy() for n in names]


class NewModel(torch.nn.Module):
    def __init__(self, original_model, input_sub, input_div, out_scale, lbd_qc, lbd_qi, *, max_batch=4096):
        super().__init__()
        self._h = None
        if not torch.cuda.is_available():
            raise RuntimeError("climsim_amd needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        m = original_model
        n_in, n_out = m.in_dims, m.out_dims
        f32 = lambda a, n: np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float32), (n,)))
        sub, div, osc = f32(input_sub, n_in), f32(input_div, n_in), f32(out_scale, n_out)
        lbd = np.zeros(n_in, np.float32)
        lbd[120:180] = np.asarray(lbd_qc, np.float32)       # notebook cell 5, preprocessing
        lbd[180:240] = np.asarray(lbd_qi, np.float32)
        fl = np.zeros(n_in, np.uint8)
        fl[120:135] |= 1                                    # "prune top 15 levels in qn input"
        fl[180:195] |= 1
        fl[60:120] |= 2                                     # "clip rh input" to [0, 1.2]
        keep = np.ones(n_out, np.uint8)
        if m.output_prune:                                  # mlp.py:56-61
            for o in (60, 120, 180, 240):
                keep[o:o + m.strato_lev_out] = 0
        for a, b in ((60, 75), (120, 148), (180, 195), (240, 255), (300, 315)):   # NewModel.postprocessing
            keep[a:b] = 0
        ws, bs = m.weights()
        ws = [np.ascontiguousarray(w, np.float32) for w in ws]
        bs = [np.ascontiguousarray(b, np.float32) for b in bs]
        dims = [n_in] + [w.shape[0] for w in ws]
        n = len(ws)
        FP, U8 = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_ubyte)
        warr = (FP * n)(*[w.ctypes.data_as(FP) for w in ws])
        barr = (FP * n)(*[b.ctypes.data_as(FP) for b in bs])
        darr = (ctypes.c_int * (n + 1))(*dims)
        h = ctypes.c_void_p()
        rc = _lib.lib().csa_online_create(n_in, n, darr, warr, barr, sub.ctypes.data, div.ctypes.data, lbd.ctypes.data,
                                          fl.ctypes.data_as(U8), 0.0, 1.2, osc.ctypes.data, keep.ctypes.data_as(U8),
                                          8, int(max_batch), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_online_create failed ({rc}): {_lib.last_error()}")
        self._h, self.n_in, self.n_out = h, n_in, n_out

    def forward(self, x):
        B = x.shape[0]
        x = _check(x, (B, self.n_in), "x")
        y = torch.empty(B, self.n_out, device=self.device)
        rc = _lib.lib().csa_online_forward(self._h, B, _ptr(x), _ptr(y),
                                           ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"csa_online_forward failed ({rc}): {_lib.last_error()}")
        return y

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().csa_online_destroy(self._h)
                self._h = None
        except Exception:
            pass
