"""Device-side mirror of the array adapters of climsim_utils/data_utils.py that sit either side of the CNN baseline
(reshape_input_for_cnn :2104-2124, reshape_target_for_cnn :2126-2150, reshape_target_from_cnn :2152-2175; V1 variable
set: two 60-level profiles + 4 input / 8 output scalars).  Same static-method names and argument meaning; tensors
are float32 on the GPU.  Everything else in that module (xarray / netCDF dataset building, plotting, evaluation
tables) is host-side I/O and outside this package."""
import ctypes

import torch

from . import _lib
from .emulator import _check, _ptr


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class data_utils:
    @staticmethod
    def _to(x, nscal, nlev=60, nprof=2):
        N = x.shape[0]
        x = _check(x, (N, nprof * nlev + nscal), "input")
        y = torch.empty(N, nlev, nprof + nscal, device=x.device)
        rc = _lib.lib().csa_cnn_reshape_to(N, nlev, nprof, nscal, _ptr(x), _ptr(y), _stream(x))
        if rc != 0:
            raise RuntimeError(f"csa_cnn_reshape_to failed ({rc}): {_lib.last_error()}")
        return y

    @staticmethod
    def reshape_input_for_cnn(npy_input, save_path=""):
        if save_path != "":
            raise NotImplementedError("file output is host-side I/O and not part of this package")
        return data_utils._to(npy_input, 4)

    @staticmethod
    def reshape_target_for_cnn(npy_target, save_path=""):
        if save_path != "":
            raise NotImplementedError("file output is host-side I/O and not part of this package")
        return data_utils._to(npy_target, 8)

    @staticmethod
    def reshape_target_from_cnn(npy_predict_cnn, save_path=""):
        if save_path != "":
            raise NotImplementedError("file output is host-side I/O and not part of this package")
        N, nlev, C = npy_predict_cnn.shape
        y = _check(npy_predict_cnn, (N, nlev, C), "npy_predict_cnn")
        x = torch.empty(N, 2 * nlev + (C - 2), device=y.device)
        rc = _lib.lib().csa_cnn_reshape_from(N, nlev, 2, C - 2, _ptr(y), _ptr(x), _stream(y))
        if rc != 0:
            raise RuntimeError(f"csa_cnn_reshape_from failed ({rc}): {_lib.last_error()}")
        return x
