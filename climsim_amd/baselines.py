"""Offline baselines of the reference re-built on the same HIP GEMM.

MLP: baseline_models/MLP/training/HPO/baseline_v1/step2_retrain/step2_retrain.py:93-121
     124 -> 768 -> 640 -> 512 -> 640 -> 640 -> 128 -> (120 linear || 8 relu), LeakyReLU(0.15).
The reference's trained weights are not in the repository (.MISSING_LARGE_BLOBS) and TensorFlow is
absent, so weights are given by the caller (PyTorch (out,in) layout = transposed Keras kernels).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .emulator import _check, _ptr

MLP_V1_UNITS = (768, 640, 512, 640, 640)     # hp_units of the published best trial (FLOP_calculation.ipynb cells 4-6)


class MLPBaseline(torch.nn.Module):
    def __init__(self, weights, biases, *, leaky_alpha=0.15, n_lin_out=120, max_batch=4096):
        super().__init__()
        self._h = None
        if not torch.cuda.is_available():
            raise RuntimeError("climsim_amd needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        ws = [np.ascontiguousarray(w, np.float32) for w in weights]
        bs = [np.ascontiguousarray(b, np.float32) for b in biases]
        dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        for a, w in zip(dims[:-1], ws):
            if w.shape[1] != a:
                raise RuntimeError("MLP weight shapes do not chain")
        n = len(ws)
        FP = ctypes.POINTER(ctypes.c_float)
        warr = (FP * n)(*[w.ctypes.data_as(FP) for w in ws])
        barr = (FP * n)(*[b.ctypes.data_as(FP) for b in bs])
        darr = (ctypes.c_int * (n + 1))(*dims)
        h = ctypes.c_void_p()
        rc = _lib.lib().csa_mlp_create(n, darr, warr, barr, float(leaky_alpha), int(n_lin_out), int(max_batch),
                                       ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_mlp_create failed ({rc}): {_lib.last_error()}")
        self._h, self.dims = h, dims

    def forward(self, x):
        B = x.shape[0]
        x = _check(x, (B, self.dims[0]), "x")
        y = torch.empty(B, self.dims[-1], device=self.device)
        rc = _lib.lib().csa_mlp_forward(self._h, B, _ptr(x), _ptr(y),
                                        ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"csa_mlp_forward failed ({rc}): {_lib.last_error()}")
        return y

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().csa_mlp_destroy(self._h)
                self._h = None
        except Exception:
            pass


class CNNBaseline(torch.nn.Module):
    """Keras ResNet-1D baseline, forward (baseline_models/CNN/training/hpo_train.py:124-200).
    weights/biases: lists in PyTorch Conv1d layout (cout,cin,k): per block conv_a, conv_b, residual 1x1; then the
    pre-output 1x1 conv (10,406,1); then the stacked Dense (10,10,1) = [Dense(2,linear); Dense(8,relu)]."""

    def __init__(self, weights, biases, *, depth=12, nlev=60, cin=6, width=406, cout=10, n_lin=2, max_batch=512):
        super().__init__()
        self._h = None
        if not torch.cuda.is_available():
            raise RuntimeError("climsim_amd needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        ws = [np.ascontiguousarray(w, np.float32) for w in weights]
        bs = [np.ascontiguousarray(b, np.float32) for b in biases]
        if len(ws) != 3 * depth + 2:
            raise RuntimeError("expected 3 weights per block plus pre-output conv and dense")
        n = len(ws)
        FP = ctypes.POINTER(ctypes.c_float)
        warr = (FP * n)(*[w.ctypes.data_as(FP) for w in ws])
        barr = (FP * n)(*[b.ctypes.data_as(FP) for b in bs])
        h = ctypes.c_void_p()
        rc = _lib.lib().csa_cnn_create(depth, nlev, cin, width, cout, n_lin, warr, barr, int(max_batch), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_cnn_create failed ({rc}): {_lib.last_error()}")
        self._h, self.nlev, self.cin, self.cout = h, nlev, cin, cout

    def forward(self, x):
        B = x.shape[0]
        x = _check(x, (B, self.nlev, self.cin), "x")
        y = torch.empty(B, self.nlev, self.cout, device=self.device)
        rc = _lib.lib().csa_cnn_forward(self._h, B, _ptr(x), _ptr(y),
                                        ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"csa_cnn_forward failed ({rc}): {_lib.last_error()}")
        return y

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().csa_cnn_destroy(self._h)
                self._h = None
        except Exception:
            pass
