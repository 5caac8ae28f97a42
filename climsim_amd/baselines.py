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


class CNNTrainer:
    """One optimiser step of the Keras CNN baseline on the HIP path (hpo_train.py:124-236: Dropout(0.175), mae_adjusted,
    keras Adam).  Columns are sharded across ranks; gradients live in ONE flat buffer, all-reduced once per step."""

    def __init__(self, weights, biases, *, depth=12, nlev=60, cin=6, width=406, cout=10, n_lin=2, dropout=0.175,
                 max_batch=512):
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
        L = _lib.lib()
        rc = L.csa_cnn_train_create(depth, nlev, cin, width, cout, n_lin, warr, barr, int(max_batch), float(dropout), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_cnn_train_create failed ({rc}): {_lib.last_error()}")
        self._h = h
        self.depth, self.nlev, self.cin, self.width, self.cout, self.dropout = depth, nlev, cin, width, cout, dropout
        self.n_params = int(L.csa_cnn_train_num_params(h))
        self.grads = torch.zeros(self.n_params, device=self.device)
        self.loss = torch.zeros(1, device=self.device)
        self.step_count = 0
        self.layers = []
        for i in range(L.csa_cnn_train_num_layers(h)):
            wo, bo = ctypes.c_long(), ctypes.c_long()
            v = [ctypes.c_int() for _ in range(5)]
            L.csa_cnn_train_layer_info(h, i, ctypes.byref(wo), ctypes.byref(bo), *[ctypes.byref(q) for q in v])
            self.layers.append((wo.value, bo.value) + tuple(q.value for q in v))

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _rc(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {_lib.last_error()}")

    def draw_masks(self, B):
        """Keep-flags of the 2*depth Dropout layers, drawn on the device in layer order."""
        if self.dropout <= 0:
            return None
        return (torch.rand(2 * self.depth, B * self.nlev, self.width, device=self.device) >= self.dropout).to(torch.uint8)

    def forward(self, x, masks=None):
        B = x.shape[0]
        x = _check(x, (B, self.nlev, self.cin), "x")
        if masks is not None:
            if masks.dtype != torch.uint8 or tuple(masks.shape) != (2 * self.depth, B * self.nlev, self.width) or not masks.is_cuda:
                raise RuntimeError("masks: expected a uint8 CUDA tensor of shape (2*depth, B*nlev, width)")
            masks = masks.contiguous()
        y = torch.empty(B, self.nlev, self.cout, device=self.device)
        mp = None if masks is None else ctypes.c_void_p(masks.data_ptr())
        self._rc(_lib.lib().csa_cnn_train_forward(self._h, B, _ptr(x), mp, _ptr(y), self._stream()), "csa_cnn_train_forward")
        return y

    def backward(self, y_true, grad_scale=1.0):
        y_true = _check(y_true, tuple(y_true.shape), "y_true")
        self._rc(_lib.lib().csa_cnn_train_backward(self._h, _ptr(y_true), float(grad_scale), _ptr(self.loss), _ptr(self.grads),
                                                   self._stream()), "csa_cnn_train_backward")
        return self.loss, self.grads

    def adam(self, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-7):
        self.step_count += 1
        self._rc(_lib.lib().csa_cnn_train_adam(self._h, _ptr(self.grads), lr, beta1, beta2, eps, self.step_count, self._stream()),
                 "csa_cnn_train_adam")

    def train_step(self, x, y_true, masks="draw", lr=1e-4, world_size=1, global_columns=None):
        """forward + loss + backward + ONE flat-gradient all-reduce (world_size > 1) + Adam; returns the loss tensor
        (this rank's share of the global mean; summed over ranks by the same all-reduce call pattern).  Ragged shards: pass
        `global_columns` so that the share is B_local / B_global instead of 1 / world_size."""
        if isinstance(masks, str):
            masks = self.draw_masks(x.shape[0])
        self.forward(x, masks)
        self.backward(y_true, (x.shape[0] / float(global_columns)) if global_columns else 1.0 / world_size)
        if world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(self.grads)
            dist.all_reduce(self.loss)
        self.adam(lr=lr)
        return self.loss

    def flat_params(self):
        t = torch.empty(self.n_params, device=self.device)
        self._rc(_lib.lib().csa_cnn_train_get_params(self._h, _ptr(t), self._stream()), "csa_cnn_train_get_params")
        return t

    def saved_activation(self, block, which, B):
        """(B, nlev, width) copy of a saved activation of the last forward (debug tap; which: 0 t1, 1 t2, 2 block output)."""
        wp = (self.width + 7) // 8 * 8
        t = torch.empty(B * self.nlev, wp, device=self.device)
        ld = ctypes.c_int()
        self._rc(_lib.lib().csa_cnn_train_get_act(self._h, block, which, _ptr(t), ctypes.byref(ld), self._stream()), "csa_cnn_train_get_act")
        return t[:, :self.width].reshape(B, self.nlev, self.width)

    def load_flat_params(self, flat):
        flat = _check(flat, (self.n_params,), "flat")
        self._rc(_lib.lib().csa_cnn_train_set_params(self._h, _ptr(flat), self._stream()), "csa_cnn_train_set_params")

    def unpack(self, flat):
        """Flat (padded GEMM layout) -> lists of Conv1d-layout weights (cout,cin,k) and biases, for tests / export."""
        flat = flat.detach().cpu()
        ws, bs = [], []
        for (wo, bo, cout, cin, k, cout_p, cin_p) in self.layers:
            w = flat[wo:wo + cout_p * k * cin_p].view(cout_p, k, cin_p)[:cout, :, :cin].permute(0, 2, 1).contiguous()
            ws.append(w)
            bs.append(flat[bo:bo + cout].clone())
        return ws, bs

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().csa_cnn_train_destroy(self._h)
                self._h = None
        except Exception:
            pass


class MLPTrainer:
    """One optimiser step of the Keras MLP baseline on the HIP path (step2_retrain.py:93-155: loss 'mse', keras Adam)."""

    def __init__(self, weights, biases, *, leaky_alpha=0.15, n_lin_out=120, max_batch=4096):
        self._h = None
        if not torch.cuda.is_available():
            raise RuntimeError("climsim_amd needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        ws = [np.ascontiguousarray(w, np.float32) for w in weights]
        bs = [np.ascontiguousarray(b, np.float32) for b in biases]
        self.dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        n = len(ws)
        FP = ctypes.POINTER(ctypes.c_float)
        warr = (FP * n)(*[w.ctypes.data_as(FP) for w in ws])
        barr = (FP * n)(*[b.ctypes.data_as(FP) for b in bs])
        darr = (ctypes.c_int * (n + 1))(*self.dims)
        h = ctypes.c_void_p()
        rc = _lib.lib().csa_mlp_train_create(n, darr, warr, barr, float(leaky_alpha), int(n_lin_out), int(max_batch), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_mlp_train_create failed ({rc}): {_lib.last_error()}")
        self._h = h
        self.n_params = int(_lib.lib().csa_mlp_train_num_params(h))
        self.grads = torch.zeros(self.n_params, device=self.device)
        self.loss = torch.zeros(1, device=self.device)
        self.step_count = 0

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _rc(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {_lib.last_error()}")

    def forward(self, x):
        B = x.shape[0]
        self._x = _check(x, (B, self.dims[0]), "x")          # kept alive: the backward reads it
        y = torch.empty(B, self.dims[-1], device=self.device)
        self._rc(_lib.lib().csa_mlp_train_forward(self._h, B, _ptr(self._x), _ptr(y), self._stream()), "csa_mlp_train_forward")
        return y

    def backward(self, y_true, grad_scale=1.0):
        y_true = _check(y_true, tuple(y_true.shape), "y_true")
        self._rc(_lib.lib().csa_mlp_train_backward(self._h, _ptr(y_true), float(grad_scale), _ptr(self.loss), _ptr(self.grads),
                                                   self._stream()), "csa_mlp_train_backward")
        return self.loss, self.grads

    def adam(self, lr=2.5e-4, beta1=0.9, beta2=0.999, eps=1e-7):
        self.step_count += 1
        self._rc(_lib.lib().csa_mlp_train_adam(self._h, _ptr(self.grads), lr, beta1, beta2, eps, self.step_count, self._stream()),
                 "csa_mlp_train_adam")

    def train_step(self, x, y_true, lr=2.5e-4, world_size=1, global_columns=None):
        self.forward(x)
        self.backward(y_true, (x.shape[0] / float(global_columns)) if global_columns else 1.0 / world_size)
        if world_size > 1:
            import torch.distributed as dist
            dist.all_reduce(self.grads)
            dist.all_reduce(self.loss)
        self.adam(lr=lr)
        return self.loss

    def flat_params(self):
        t = torch.empty(self.n_params, device=self.device)
        self._rc(_lib.lib().csa_mlp_train_copy_params(self._h, 0, _ptr(t), self._stream()), "csa_mlp_train_copy_params")
        return t

    def unpack(self, flat):
        flat = flat.detach().cpu()
        ws, bs, o = [], [], 0
        for i in range(len(self.dims) - 1):
            n = self.dims[i + 1] * self.dims[i]
            ws.append(flat[o:o + n].view(self.dims[i + 1], self.dims[i]).clone()); o += n
            bs.append(flat[o:o + self.dims[i + 1]].clone()); o += self.dims[i + 1]
        return ws, bs

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().csa_mlp_train_destroy(self._h)
                self._h = None
        except Exception:
            pass
