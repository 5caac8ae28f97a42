"""Host mirrors of the reference's stochastic recurrent layers (rnn/models_torch_kernels.py), same class
names and forward signatures; the noise eps ~ N(0,1) the reference draws inside forward can be passed
explicitly (eps=...) and is drawn on the GPU otherwise.  All arithmetic runs in stoch.hip behind the C ABI."""
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


class _StochBase(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self._h = None
        if not torch.cuda.is_available():
            raise RuntimeError("climsim_amd needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().csa_stoch_destroy(self._h)
                self._h = None
        except Exception:
            pass


class MyStochasticGRULayer5(_StochBase):
    """forward(input_seq (T,B,nx), hidden (B,H)) -> outputs (T,B,H)   (models_torch_kernels.py:834-891)"""

    def __init__(self, weight_ih, weight_zh, weight_encoder, bias_ih=None, bias_zh=None, max_rows=60 * 4096):
        super().__init__()
        k = [_host(a) for a in (weight_ih, weight_zh, weight_encoder, bias_ih, bias_zh)]
        self.input_size, self.hidden_size = k[0][0].shape[0], k[1][0].shape[0]
        h = ctypes.c_void_p()
        rc = _lib.lib().csa_stoch_gru5_create(self.input_size, self.hidden_size, k[0][1], k[1][1], k[2][1], k[3][1], k[4][1],
                                              int(max_rows), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_stoch_gru5_create failed ({rc}): {_lib.last_error()}")
        self._h = h

    def forward(self, input_seq, hidden, eps=None):
        T, B, _ = input_seq.shape
        H = self.hidden_size
        x = _check(input_seq, (T, B, self.input_size), "input_seq")
        h0 = _check(hidden, (B, H), "hidden")
        eps = torch.randn(T, B, H, device=self.device) if eps is None else _check(eps, (T, B, H), "eps")
        out = torch.empty(T, B, H, device=self.device)
        rc = _lib.lib().csa_stoch_gru5_forward(self._h, T, B, _ptr(x), _ptr(h0), _ptr(eps), _ptr(out), self._stream())
        if rc != 0:
            raise RuntimeError(f"csa_stoch_gru5_forward failed ({rc}): {_lib.last_error()}")
        return out


class MyStochasticLSTMLayer4(_StochBase):
    """forward(input_seq (T,B,nx), (hx, cx)) -> (outputs (T,B,H), (hx, cx))   (models_torch_kernels.py:1474-1531)"""

    def __init__(self, weight_encoder, hidden_size, max_rows=60 * 4096):
        super().__init__()
        w, wp = _host(weight_encoder)
        self.hidden_size = int(hidden_size)
        self.input_size = w.shape[0] - self.hidden_size
        if w.shape[1] != 5 * self.hidden_size:
            raise RuntimeError("weight_encoder must be (input_size + hidden_size, 5*hidden_size)")
        h = ctypes.c_void_p()
        rc = _lib.lib().csa_stoch_lstm4_create(self.input_size, self.hidden_size, wp, int(max_rows), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_stoch_lstm4_create failed ({rc}): {_lib.last_error()}")
        self._h = h

    def forward(self, input_seq, state, eps=None):
        T, B, _ = input_seq.shape
        H = self.hidden_size
        x = _check(input_seq, (T, B, self.input_size), "input_seq")
        h0, c0 = _check(state[0], (B, H), "hx"), _check(state[1], (B, H), "cx")
        eps = torch.randn(T, B, H, device=self.device) if eps is None else _check(eps, (T, B, H), "eps")
        out = torch.empty(T, B, H, device=self.device)
        hT, cT = torch.empty(B, H, device=self.device), torch.empty(B, H, device=self.device)
        rc = _lib.lib().csa_stoch_lstm4_forward(self._h, T, B, _ptr(x), _ptr(h0), _ptr(c0), _ptr(eps), _ptr(out), _ptr(hT),
                                                _ptr(cT), self._stream())
        if rc != 0:
            raise RuntimeError(f"csa_stoch_lstm4_forward failed ({rc}): {_lib.last_error()}")
        return out, (hT, cT)
