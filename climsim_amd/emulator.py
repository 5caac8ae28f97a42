"""Emulator: owns one csa_emulator handle (weights + scratch on the current HIP device).

PyTorch is used for device memory and streams only; every FLOP of the path runs in the
hand-written HIP kernels behind the C ABI.
"""
import ctypes

import numpy as np
import torch

from . import _lib

# state_dict key (rnn/models/models.py) -> csa_params field
STATE_DICT_MAP = {
    "mlp_initial.weight": "mlp_initial_w", "mlp_initial.bias": "mlp_initial_b",
    "mlp_surface1.weight": "mlp_surface1_w", "mlp_surface1.bias": "mlp_surface1_b",
    "mlp_surface2.weight": "mlp_surface2_w", "mlp_surface2.bias": "mlp_surface2_b",
    "mlp_toa1.weight": "mlp_toa1_w", "mlp_toa1.bias": "mlp_toa1_b",
    "mlp_toa2.weight": "mlp_toa2_w", "mlp_toa2.bias": "mlp_toa2_b",
    "rnn1.weight_ih_l0": "rnn1_w_ih", "rnn1.weight_hh_l0": "rnn1_w_hh",
    "rnn1.bias_ih_l0": "rnn1_b_ih", "rnn1.bias_hh_l0": "rnn1_b_hh",
    "rnn2.weight_ih_l0": "rnn2_w_ih", "rnn2.weight_hh_l0": "rnn2_w_hh",
    "rnn2.bias_ih_l0": "rnn2_b_ih", "rnn2.bias_hh_l0": "rnn2_b_hh",
    "rnn0.weight_ih_l0": "rnn0_w_ih", "rnn0.weight_hh_l0": "rnn0_w_hh",
    "rnn0.bias_ih_l0": "rnn0_b_ih", "rnn0.bias_hh_l0": "rnn0_b_hh",
    "rnn2.weight_encoder": "rnn2_weight_encoder",
    "mlp_latent.weight": "mlp_latent_w", "mlp_latent.bias": "mlp_latent_b",
    "mlp_output.weight": "mlp_output_w", "mlp_output.bias": "mlp_output_b",
    "mlp_surface_output.weight": "mlp_surface_output_w", "mlp_surface_output.bias": "mlp_surface_output_b",
}
CONST_KEYS = ["xmean_lev", "xdiv_lev", "xmean_sca", "xdiv_sca", "lbd_qc", "lbd_qi",
              "yscale_lev", "yscale_sca", "hyam", "hybm"]


def _np32(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _check(t, shape, name):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA/HIP tensor, got {t.device} (climsim_amd has no CPU path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Emulator:
    def __init__(self, consts, state_dict, *, legacy, use_lstm=True, mp_mode=1, output_prune=False,
                 snowhice_fix=False, qinput_prune=False, rh_prune=False, scrub_inf=False,
                 scrub_out_nan=False, q_input_mode=0, v5_input=False, max_batch=4096, device=None):
        self._h = None
        L = _lib.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("climsim_amd needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._host = {}
        params = _lib.CsaParams()
        for k in CONST_KEYS:
            self._host[k] = _np32(consts[k])
            setattr(params, k, self._host[k].ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        if v5_input:
            if "lbd_qn" not in consts:
                raise RuntimeError("v5_input needs consts['lbd_qn'] (models.py:171)")
            self._host["lbd_qn"] = _np32(consts["lbd_qn"])
            params.lbd_qn = self._host["lbd_qn"].ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        for sd, f in STATE_DICT_MAP.items():
            if sd in state_dict:
                self._host[f] = _np32(state_dict[sd])
                setattr(params, f, self._host[f].ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        w = self._host
        cfg = _lib.CsaConfig()
        cfg.nlev, cfg.nx = w["xmean_lev"].shape
        cfg.nx_sfc = w["xmean_sca"].shape[0]
        cfg.ny = w["mlp_output_w"].shape[0]
        cfg.ny_sfc = w["mlp_surface_output_w"].shape[0]
        # add_stochastic_layer (models.py:405-412) is recognised by its parameters: rnn0.* + rnn2.weight_encoder
        stochastic = "rnn2_weight_encoder" in w
        cfg.add_stochastic_layer = int(stochastic)
        cfg.nh1 = w["rnn1_w_hh"].shape[1]
        cfg.nh2 = w["rnn2_weight_encoder"].shape[1] // 5 if stochastic else w["rnn2_w_hh"].shape[1]
        cfg.nh_mem = w["mlp_latent_w"].shape[0] if "mlp_latent_w" in w else 0
        G = 4 if use_lstm else 3
        first = "rnn0_w_ih" if stochastic else "rnn1_w_ih"
        if w["rnn1_w_hh"].shape[0] != G * cfg.nh1 or w[first].shape[1] != cfg.nh1 + cfg.nh_mem:
            raise RuntimeError("state_dict shapes inconsistent with use_lstm / nh_mem")
        cfg.use_lstm, cfg.legacy = int(use_lstm), int(legacy)
        cfg.output_prune, cfg.mp_mode = int(output_prune), int(mp_mode)
        cfg.snowhice_fix, cfg.qinput_prune, cfg.rh_prune = int(snowhice_fix), int(qinput_prune), int(rh_prune)
        cfg.scrub_inf, cfg.scrub_out_nan = int(scrub_inf), int(scrub_out_nan)
        cfg.q_input_mode = int(q_input_mode)
        cfg.v5_input = int(bool(v5_input))
        self.cfg = cfg
        self.max_batch = int(max_batch)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = L.csa_create(ctypes.byref(cfg), ctypes.byref(params), self.max_batch, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_create failed ({rc}): {_lib.last_error()}")
        self._h = h

    def load_state_dict(self, state_dict):
        """Replace the weights of this handle (same shapes): load_state_dict of the reference module.  Constants
        (normalisation, grid) are kept."""
        params = _lib.CsaParams()
        for k in CONST_KEYS + (["lbd_qn"] if "lbd_qn" in self._host else []):
            setattr(params, k, self._host[k].ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        for sd, f in STATE_DICT_MAP.items():
            if sd in state_dict:
                a = _np32(state_dict[sd])
                if f in self._host and a.shape != self._host[f].shape:
                    raise RuntimeError(f"{sd}: shape {a.shape} does not match the handle's {self._host[f].shape}")
                self._host[f] = a
                setattr(params, f, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        with torch.cuda.device(self.device):
            rc = _lib.lib().csa_set_params(self._h, ctypes.byref(params))
        if rc != 0:
            raise RuntimeError(f"csa_set_params failed ({rc}): {_lib.last_error()}")

    def close(self):
        if self._h is not None:
            _lib.lib().csa_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers ---------------------------------------------------------------------------
    @property
    def packed_width(self):
        c = self.cfg
        return 6 * c.nlev + c.ny_sfc + c.nlev * c.nh_mem

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _noise(self, B, hx2, cx2):
        c = self.cfg
        if not c.legacy:
            return None, None
        # the legacy artefacts draw hx2 then cx2 = randn(B, nh) inside forward
        if hx2 is None:
            hx2 = torch.randn(B, c.nh2, device=self.device)
        if cx2 is None and c.use_lstm:
            cx2 = torch.randn(B, c.nh2, device=self.device)
        hx2 = _check(hx2, (B, c.nh2), "hx2")
        cx2 = _check(cx2, (B, c.nh2), "cx2") if c.use_lstm else None
        return hx2, cx2

    def _rc(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {_lib.last_error()}")

    # ---- calls ----------------------------------------------------------------------------------
    def forward_packed(self, x_main, x_sfc, rnn1_mem=None, hx2=None, cx2=None, out=None):
        c = self.cfg
        B = x_main.shape[0]
        x_main = _check(x_main, (B, c.nlev, c.nx), "x_main")
        x_sfc = _check(x_sfc, (B, c.nx_sfc), "x_sfc")
        if c.nh_mem > 0:
            if rnn1_mem is None:
                raise RuntimeError("rnn1_mem is required by a model with memory")
            rnn1_mem = _check(rnn1_mem, (B, c.nlev, c.nh_mem), "rnn1_mem")
        hx2, cx2 = self._noise(B, hx2, cx2)
        y = torch.empty(B, self.packed_width, device=self.device) if out is None else \
            _check(out, (B, self.packed_width), "out")
        rc = _lib.lib().csa_forward_packed(self._h, B, _ptr(x_main), _ptr(x_sfc), _ptr(rnn1_mem),
                                           _ptr(hx2), _ptr(cx2), _ptr(y), self._stream())
        self._rc(rc, "csa_forward_packed")
        return y

    def _stoch_noise(self, B, noise):
        """(hx0, cx0, eps): drawn here in the reference's order (models.py:466-468, models_torch_kernels.py:1497)
        when not given, so a seeded torch generator reproduces a run."""
        c = self.cfg
        if noise is None:
            noise = (torch.randn(B, c.nh1, device=self.device), torch.randn(B, c.nh1, device=self.device),
                     torch.randn(c.nlev, B, c.nh2, device=self.device))
        hx0, cx0, eps = noise
        return _check(hx0, (B, c.nh1), "hx0"), _check(cx0, (B, c.nh1), "cx0"), _check(eps, (c.nlev, B, c.nh2), "eps")

    def forward_tuple(self, x_main, x_sfc, rnn1_mem, noise=None):
        c = self.cfg
        B = x_main.shape[0]
        x_main = _check(x_main, (B, c.nlev, c.nx - (1 if c.q_input_mode == 1 else 0)), "x_main")
        x_sfc = _check(x_sfc, (B, c.nx_sfc), "x_sfc")
        rnn1_mem = _check(rnn1_mem, (c.nlev, B, c.nh_mem), "rnn1_mem")
        nyo = 6 if c.mp_mode != 0 else c.ny
        out_lev = torch.empty(B, c.nlev, nyo, device=self.device)
        out_sfc = torch.empty(B, c.ny_sfc, device=self.device)
        mem_out = torch.empty(c.nlev, B, c.nh_mem, device=self.device)
        if c.add_stochastic_layer:
            hx0, cx0, eps = self._stoch_noise(B, noise)
            rc = _lib.lib().csa_forward_tuple_noise(self._h, B, _ptr(x_main), _ptr(x_sfc), _ptr(rnn1_mem), _ptr(hx0),
                                                    _ptr(cx0), _ptr(eps), _ptr(out_lev), _ptr(out_sfc), _ptr(mem_out),
                                                    self._stream())
            self._rc(rc, "csa_forward_tuple_noise")
            return out_lev, out_sfc, mem_out
        rc = _lib.lib().csa_forward_tuple(self._h, B, _ptr(x_main), _ptr(x_sfc), _ptr(rnn1_mem),
                                          _ptr(out_lev), _ptr(out_sfc), _ptr(mem_out), self._stream())
        self._rc(rc, "csa_forward_tuple")
        return out_lev, out_sfc, mem_out

    def model_forward(self, x_main_n, x_sfc_n, rnn_mem=None, hx2=None, cx2=None, noise=None):
        c = self.cfg
        B = x_main_n.shape[0]
        x_main_n = _check(x_main_n, (B, c.nlev, c.nx), "x_main")
        x_sfc_n = _check(x_sfc_n, (B, c.nx_sfc), "x_sfc")
        mem_shape = (B, c.nlev, c.nh_mem) if c.legacy else (c.nlev, B, c.nh_mem)
        mem_out = None
        if c.nh_mem > 0:
            rnn_mem = _check(rnn_mem, mem_shape, "rnn_mem")
            mem_out = torch.empty(mem_shape, device=self.device)
        out = torch.empty(B, c.nlev, c.ny, device=self.device)
        out_sfc = torch.empty(B, c.ny_sfc, device=self.device)
        if c.add_stochastic_layer:
            hx0, cx0, eps = self._stoch_noise(B, noise)
            rc = _lib.lib().csa_model_forward_noise(self._h, B, _ptr(x_main_n), _ptr(x_sfc_n), _ptr(rnn_mem), _ptr(hx0),
                                                    _ptr(cx0), _ptr(eps), _ptr(out), _ptr(out_sfc), _ptr(mem_out),
                                                    self._stream())
            self._rc(rc, "csa_model_forward_noise")
            return out, out_sfc, mem_out
        hx2, cx2 = self._noise(B, hx2, cx2)
        rc = _lib.lib().csa_model_forward(self._h, B, _ptr(x_main_n), _ptr(x_sfc_n), _ptr(rnn_mem),
                                          _ptr(hx2), _ptr(cx2), _ptr(out), _ptr(out_sfc), _ptr(mem_out),
                                          self._stream())
        self._rc(rc, "csa_model_forward")
        return out, out_sfc, mem_out

    def set_halves(self, enable):
        """Two column halves on two streams (legacy models, B >= 64); bit-identical results.
        True / False force it, None restores the default (automatic from 640 columns)."""
        return _lib.lib().csa_set_halves(self._h, 2 if enable is None else int(bool(enable)))

    @staticmethod
    def set_gemm_split(enable):
        """Process-wide opt-in (csa_set_gemm_split): the input projections of calls above the small-GEMM threshold run with every fp32
        operand split exactly into three bf16 values, six partial products on the bf16 matrix pipe and fp32 accumulation.
        Default off: the fp32 MFMA chain."""
        return _lib.lib().csa_set_gemm_split(int(bool(enable)))

    def set_rec1_max_batch(self, max_batch):
        """Largest batch that uses the one-column-per-workgroup recurrent kernel (default 256); 0 disables it."""
        self._rc(_lib.lib().csa_set_rec1_max_batch(self._h, int(max_batch)), "csa_set_rec1_max_batch")

    def forward_packed_noise(self, x_main, x_sfc, rnn1_mem, eps_prev, hx0=None, cx0=None):
        """Stateful + AR-noise packed wrapper (save_wrapper_mem.py:682-727) around the stochastic model; all state batch-first."""
        c = self.cfg
        B = x_main.shape[0]
        x_main = _check(x_main, (B, c.nlev, c.nx), "x_main")
        x_sfc = _check(x_sfc, (B, c.nx_sfc), "x_sfc")
        rnn1_mem = _check(rnn1_mem, (B, c.nlev, c.nh_mem), "rnn1_mem")
        eps_prev = _check(eps_prev, (B, c.nlev, c.nh2), "eps_prev")
        hx0 = torch.randn(B, c.nh1, device=self.device) if hx0 is None else _check(hx0, (B, c.nh1), "hx0")
        cx0 = torch.randn(B, c.nh1, device=self.device) if cx0 is None else _check(cx0, (B, c.nh1), "cx0")
        y = torch.empty(B, 6 * c.nlev + c.ny_sfc + c.nlev * (c.nh_mem + c.nh2), device=self.device)
        self._rc(_lib.lib().csa_forward_packed_noise(self._h, B, _ptr(x_main), _ptr(x_sfc), _ptr(rnn1_mem), _ptr(hx0), _ptr(cx0),
                                                     _ptr(eps_prev), _ptr(y), self._stream()), "csa_forward_packed_noise")
        return y

    def postprocess(self, out, out_sfc, x_denorm):
        """RNN_autoreg.postprocessing (models.py:273-339): (B,nlev,ny), (B,ny_sfc), raw (B,nlev,>=4) -> (B,nlev,6), (B,ny_sfc)."""
        c = self.cfg
        if c.mp_mode == 0:            # models.py:278-279: nothing is done
            return out, out_sfc
        B = out.shape[0]
        out = _check(out, (B, c.nlev, c.ny), "out")
        out_sfc = _check(out_sfc, (B, c.ny_sfc), "out_sfc")
        if x_denorm.dim() != 3 or x_denorm.shape[:2] != (B, c.nlev) or x_denorm.shape[2] < 4:
            raise RuntimeError(f"x_denorm: expected shape ({B}, {c.nlev}, >=4), got {tuple(x_denorm.shape)}")
        x_denorm = _check(x_denorm, tuple(x_denorm.shape), "x_denorm")
        o6 = torch.empty(B, c.nlev, 6, device=self.device)
        osd = torch.empty(B, c.ny_sfc, device=self.device)
        self._rc(_lib.lib().csa_postprocess(self._h, B, _ptr(out), _ptr(out_sfc), _ptr(x_denorm), int(x_denorm.shape[2]),
                                            _ptr(o6), _ptr(osd), self._stream()), "csa_postprocess")
        return o6, osd

    def debug_stage(self, stage, B, ins, out_shapes):
        """Run ONE launch of the forward path on given device inputs (csa_debug_stage; internal layouts, see the header)."""
        ins = list(ins) + [None] * (5 - len(ins))
        outs = [torch.empty(sh, device=self.device) for sh in out_shapes] + [None] * (2 - len(out_shapes))
        self._rc(_lib.lib().csa_debug_stage(self._h, int(stage), int(B), *[_ptr(t) for t in ins], _ptr(outs[0]), _ptr(outs[1]),
                                            self._stream()), "csa_debug_stage")
        return outs[:len(out_shapes)]

    def set_profiling(self, enable):
        self._rc(_lib.lib().csa_set_profiling(self._h, int(bool(enable))), "csa_set_profiling")

    def reset_profile(self):
        self._rc(_lib.lib().csa_reset_profile(self._h), "csa_reset_profile")

    def get_profile(self):
        """{stage: average ms per launch} over the profiled calls since the last reset, and the call count."""
        arr = (ctypes.c_double * 6)()
        n = ctypes.c_long()
        self._rc(_lib.lib().csa_get_profile(self._h, arr, 6, ctypes.byref(n)), "csa_get_profile")
        names = [_lib.lib().csa_stage_name(i).decode() for i in range(6)]
        return dict(zip(names, list(arr))), n.value

    def taps(self, B):
        """(rnn1out, rnn2out) of the last call as (nlev,B,nh) tensors (device-to-device copies)."""
        c = self.cfg
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        outs = []
        torch.cuda.synchronize(self.device)
        for fn, nh in ((_lib.lib().csa_tap_rnn1, c.nh1), (_lib.lib().csa_tap_rnn2, c.nh2)):
            t = torch.empty(c.nlev, B, nh, device=self.device)
            rc = hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(fn(self._h)), t.numel() * 4, 3)
            if rc != 0:
                raise RuntimeError(f"hipMemcpy failed ({rc})")
            outs.append(t)
        return outs
