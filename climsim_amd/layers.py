"""Host mirrors of the reference's stochastic recurrent layers (rnn/models_torch_kernels.py), same class
names and forward signatures; the noise eps ~ N(0,1) the reference draws inside forward can be passed
explicitly (eps=...) and is drawn on the GPU otherwise.  All arithmetic runs in stoch.hip / stoch_bwd.hip behind the C ABI.

Like the reference's GPU path (FusedCUDAStochasticGRUSequence, :795-841) the layers are differentiable through a
torch.autograd.Function whose backward is native code: with `requires_grad` inputs or `train()` mode the forward keeps the
activations BPTT needs and `.backward()` fills the `.grad` of the layer's parameters (reference names and (in, out) layouts),
of the input sequence and of the initial state.  The kernels' packed weight copies follow in-place parameter updates by themselves
(tensor version counters, checked per forward); `sync_params()` forces the re-pack."""
import ctypes

import numpy as np
import torch

from . import _lib
from .emulator import _check, _ptr

_FP = ctypes.POINTER(ctypes.c_float)


def _host(a):
    if a is None:
        return None, None
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(_FP)


class _StochFn(torch.autograd.Function):
    """forward(x, h0, c0 | None, eps, layer, *params) -> out [, hT, cT]; backward = csa_stoch_*_backward.

    Every forward owns the activations its backward will consume (one torch buffer, installed in the native handle for the two
    calls only), as the reference's autograd function keeps them in ctx (models_torch_kernels.py:795-841): several forwards may be
    pending, inference calls in between do not disturb them.  The backward overwrites them in place, so a second backward through
    the same forward (retain_graph) raises instead of returning wrong gradients, and so does a backward after sync_params()
    re-packed different weights."""

    @staticmethod
    def forward(ctx, x, h0, c0, eps, layer, *params):
        ctx.layer, ctx.shape = layer, x.shape
        T, B, _ = x.shape
        layer._enable_training()
        L = _lib.lib()
        ctx.acts = torch.empty(L.csa_stoch_activation_floats(layer._h, T, B), device=x.device)
        ctx.generation, ctx.consumed = layer._generation, False
        layer._set_acts(ctx.acts, T, B)
        try:
            out, hT, cT = layer._run(x, h0, c0, eps, train=True)
        finally:
            layer._set_acts(None, T, B)
        ctx.save_for_backward(x, eps)
        ctx.lstm = c0 is not None
        return (out, hT, cT) if ctx.lstm else out

    @staticmethod
    def backward(ctx, d_out, d_hT=None, d_cT=None):
        layer = ctx.layer
        if ctx.consumed:
            raise RuntimeError("the saved activations of this stochastic-layer forward were consumed by an earlier backward "
                               "(the native BPTT overwrites them in place): call forward again instead of retain_graph")
        if ctx.generation != layer._generation:
            raise RuntimeError("sync_params() re-packed the layer's weights between this forward and its backward")
        x, eps = ctx.saved_tensors
        T, B, _ = ctx.shape
        H, L = layer.hidden_size, _lib.lib()
        d_out = d_out.contiguous()
        d_x = torch.empty_like(x)
        d_h0 = torch.empty(B, H, device=x.device)
        d_eps = torch.empty_like(eps)
        flat = torch.zeros(L.csa_stoch_num_params(layer._h), device=x.device)
        ctx.consumed = True
        layer._set_acts(ctx.acts, T, B)
        try:
            if ctx.lstm:
                d_c0 = torch.empty(B, H, device=x.device)
                rc = L.csa_stoch_lstm4_backward(layer._h, T, B, _ptr(x), _ptr(eps), _ptr(d_out),
                                                _ptr(None if d_hT is None else d_hT.contiguous()),
                                                _ptr(None if d_cT is None else d_cT.contiguous()), _ptr(d_x), _ptr(d_h0), _ptr(d_c0),
                                                _ptr(d_eps), _ptr(flat), layer._stream())
            else:
                d_c0 = None
                rc = L.csa_stoch_gru5_backward(layer._h, T, B, _ptr(x), _ptr(eps), _ptr(d_out), _ptr(d_x), _ptr(d_h0), _ptr(d_eps),
                                               _ptr(flat), layer._stream())
        finally:
            layer._set_acts(None, T, B)
        if rc != 0:
            raise RuntimeError(f"csa_stoch backward failed ({rc}): {_lib.last_error()}")
        ctx.acts.record_stream(torch.cuda.current_stream(x.device))
        ctx.acts = None
        gp, off = [], 0
        for p in layer._param_list():
            gp.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return (d_x, d_h0, d_c0, d_eps, None, *gp)


class _StochBase(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self._h = None
        self._training_enabled = False
        self._generation = 0
        if not torch.cuda.is_available():
            raise RuntimeError("climsim_amd needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _enable_training(self):
        if not self._training_enabled:
            rc = _lib.lib().csa_stoch_enable_training(self._h)
            if rc != 0:
                raise RuntimeError(f"csa_stoch_enable_training failed ({rc}): {_lib.last_error()}")
            self._training_enabled = True

    def _set_acts(self, acts, T, B):
        rc = _lib.lib().csa_stoch_set_activations(self._h, _ptr(acts), int(T), int(B))
        if rc != 0:
            raise RuntimeError(f"csa_stoch_set_activations failed ({rc}): {_lib.last_error()}")

    def _versions(self):
        return tuple((id(p), p._version) for p in self._param_list())

    def _refresh(self):
        """The kernels run on packed copies of the parameters: an in-place update (optimizer.step(), load_state_dict(), copy_)
        bumps the tensors' version counters, and the copies are re-packed before the next forward (sync_params() does it by hand)."""
        if self._versions() != self._packed_versions:
            self.sync_params()

    def _wants_grad(self, *tensors):
        return torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in tensors)
                                            or any(p.requires_grad for p in self._param_list()))

    def sync_params(self):
        """Re-create the native handle from the current parameter values (after an optimiser step)."""
        if self._h is not None:
            _lib.lib().csa_stoch_destroy(self._h)
            self._h = None
        self._training_enabled = False
        self._generation += 1
        self._create()
        self._packed_versions = self._versions()

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().csa_stoch_destroy(self._h)
                self._h = None
        except Exception:
            pass


class MyStochasticGRULayer5(_StochBase):
    """forward(input_seq (T,B,nx), hidden (B,H)) -> outputs (T,B,H)   (models_torch_kernels.py:834-891)"""

    def __init__(self, weight_ih, weight_zh, weight_encoder, bias_ih=None, bias_zh=None, max_rows=60 * 4096, requires_grad=False):
        super().__init__()
        mk = lambda a: None if a is None else torch.nn.Parameter(torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a),
                                                                                 dtype=torch.float32).to(self.device), requires_grad=requires_grad)
        self.weight_ih, self.weight_zh, self.weight_encoder = mk(weight_ih), mk(weight_zh), mk(weight_encoder)
        self.bias_ih, self.bias_zh = mk(bias_ih), mk(bias_zh)
        self.use_bias = bias_ih is not None and bias_zh is not None
        self.input_size, self.hidden_size = self.weight_ih.shape[0], self.weight_zh.shape[0]
        self.max_rows = int(max_rows)
        self._create()
        self._packed_versions = self._versions()

    def _param_list(self):
        return [self.weight_ih, self.weight_zh, self.weight_encoder] + ([self.bias_ih, self.bias_zh] if self.use_bias else [])

    def _create(self):
        k = [_host(a) for a in (self.weight_ih, self.weight_zh, self.weight_encoder, self.bias_ih, self.bias_zh)]
        self._keep = k
        h = ctypes.c_void_p()
        rc = _lib.lib().csa_stoch_gru5_create(self.input_size, self.hidden_size, k[0][1], k[1][1], k[2][1], k[3][1], k[4][1],
                                              self.max_rows, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_stoch_gru5_create failed ({rc}): {_lib.last_error()}")
        self._h = h

    def _run(self, x, h0, c0, eps, train):
        T, B, _ = x.shape
        out = torch.empty(T, B, self.hidden_size, device=self.device)
        if train:
            self._enable_training()
        fn = _lib.lib().csa_stoch_gru5_forward_train if train else _lib.lib().csa_stoch_gru5_forward
        rc = fn(self._h, T, B, _ptr(x), _ptr(h0), _ptr(eps), _ptr(out), self._stream())
        if rc != 0:
            raise RuntimeError(f"csa_stoch_gru5_forward failed ({rc}): {_lib.last_error()}")
        return out, None, None

    def forward(self, input_seq, hidden, eps=None):
        T, B, _ = input_seq.shape
        H = self.hidden_size
        x = _check(input_seq, (T, B, self.input_size), "input_seq")
        h0 = _check(hidden, (B, H), "hidden")
        eps = torch.randn(T, B, H, device=self.device) if eps is None else _check(eps, (T, B, H), "eps")
        self._refresh()
        if self._wants_grad(x, h0):
            return _StochFn.apply(x, h0, None, eps, self, *self._param_list())
        return self._run(x, h0, None, eps, train=False)[0]


class MyStochasticLSTMLayer4(_StochBase):
    """forward(input_seq (T,B,nx), (hx, cx)) -> (outputs (T,B,H), (hx, cx))   (models_torch_kernels.py:1474-1531)"""

    def __init__(self, weight_encoder, hidden_size, max_rows=60 * 4096, requires_grad=False):
        super().__init__()
        w = torch.as_tensor(np.asarray(weight_encoder.detach().cpu() if isinstance(weight_encoder, torch.Tensor) else weight_encoder),
                            dtype=torch.float32)
        self.hidden_size = int(hidden_size)
        self.input_size = w.shape[0] - self.hidden_size
        if w.shape[1] != 5 * self.hidden_size:
            raise RuntimeError("weight_encoder must be (input_size + hidden_size, 5*hidden_size)")
        self.weight_encoder = torch.nn.Parameter(w.to(self.device), requires_grad=requires_grad)
        self.max_rows = int(max_rows)
        self._create()
        self._packed_versions = self._versions()

    def _param_list(self):
        return [self.weight_encoder]

    def _create(self):
        self._keep = _host(self.weight_encoder)
        h = ctypes.c_void_p()
        rc = _lib.lib().csa_stoch_lstm4_create(self.input_size, self.hidden_size, self._keep[1], self.max_rows, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_stoch_lstm4_create failed ({rc}): {_lib.last_error()}")
        self._h = h

    def _run(self, x, h0, c0, eps, train):
        T, B, _ = x.shape
        H = self.hidden_size
        out = torch.empty(T, B, H, device=self.device)
        hT, cT = torch.empty(B, H, device=self.device), torch.empty(B, H, device=self.device)
        if train:
            self._enable_training()
        fn = _lib.lib().csa_stoch_lstm4_forward_train if train else _lib.lib().csa_stoch_lstm4_forward
        rc = fn(self._h, T, B, _ptr(x), _ptr(h0), _ptr(c0), _ptr(eps), _ptr(out), _ptr(hT), _ptr(cT), self._stream())
        if rc != 0:
            raise RuntimeError(f"csa_stoch_lstm4_forward failed ({rc}): {_lib.last_error()}")
        return out, hT, cT

    def forward(self, input_seq, state, eps=None):
        T, B, _ = input_seq.shape
        H = self.hidden_size
        x = _check(input_seq, (T, B, self.input_size), "input_seq")
        h0, c0 = _check(state[0], (B, H), "hx"), _check(state[1], (B, H), "cx")
        eps = torch.randn(T, B, H, device=self.device) if eps is None else _check(eps, (T, B, H), "eps")
        self._refresh()
        if self._wants_grad(x, h0, c0):
            out, hT, cT = _StochFn.apply(x, h0, c0, eps, self, *self._param_list())
        else:
            out, hT, cT = self._run(x, h0, c0, eps, train=False)
        return out, (hT, cT)
