"""Device-side mirror of the array adapters of climsim_utils/data_utils.py that sit either side of the CNN baseline
(reshape_input_for_cnn :2104-2124, reshape_target_for_cnn :2126-2150, reshape_target_from_cnn :2152-2175; V1 variable
set: two 60-level profiles + 4 input / 8 output scalars).  Same static-method names and argument meaning; tensors
are float32 on the GPU.  Also the evaluation scores calc_MAE / calc_RMSE / calc_R2 / calc_bias / calc_CRPS (:1843-1935): same
method names and `avg_grid` meaning, (time, grid, level) or (time, grid) tensors, reductions in csrc/evalm.hip.
Everything else in that module (xarray / netCDF dataset building, plotting, DataFrame tables) is host-side I/O and
outside this package."""
import ctypes

import torch

from . import _lib
from .emulator import _check, _ptr


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class data_utils:
    def __init__(self, num_latlon=384):
        self.num_latlon = num_latlon

    # ---- evaluation scores (climsim_utils/data_utils.py:1843-1935) ----
    def _scores(self, pred, target):
        assert pred.shape[1] == self.num_latlon
        assert pred.shape == target.shape
        T, G = pred.shape[0], pred.shape[1]
        L = pred.shape[2] if pred.dim() == 3 else 1
        pred, target = _check(pred, tuple(pred.shape), "pred"), _check(target, tuple(target.shape), "target")
        return T, G, L, pred, target

    def _metrics(self, pred, target, avg_grid):
        T, G, L, p, t = self._scores(pred, target)
        nb = _lib.lib().csa_eval_scratch_bytes(T, G, L, 0)
        scratch = torch.empty(nb, dtype=torch.uint8, device=p.device)
        out = torch.empty((4, L) if avg_grid else (4, G, L), device=p.device)
        rc = _lib.lib().csa_eval_metrics(T, G, L, _ptr(p), _ptr(t), int(bool(avg_grid)),
                                         ctypes.c_void_p(scratch.data_ptr()), _ptr(out), _stream(p))
        if rc != 0:
            raise RuntimeError(f"csa_eval_metrics failed ({rc}): {_lib.last_error()}")
        if pred.dim() == 2:                      # scalars: the reference returns () or (grid,)
            out = out.reshape(4) if avg_grid else out.reshape(4, G)
        return out

    def calc_all(self, pred, target, avg_grid=True):
        """The four scores of one (pred, target) pair from a single pass over the data: dict MAE / RMSE / R2 / bias."""
        out = self._metrics(pred, target, avg_grid)
        return {"MAE": out[0], "RMSE": out[1], "R2": out[2], "bias": out[3]}

    def calc_MAE(self, pred, target, avg_grid=True):
        return self._metrics(pred, target, avg_grid)[0]

    def calc_RMSE(self, pred, target, avg_grid=True):
        return self._metrics(pred, target, avg_grid)[1]

    def calc_R2(self, pred, target, avg_grid=True):
        return self._metrics(pred, target, avg_grid)[2]

    def calc_bias(self, pred, target, avg_grid=True):
        return self._metrics(pred, target, avg_grid)[3]

    def calc_CRPS(self, samplepreds, target, avg_grid=True):
        assert samplepreds.shape[1] == self.num_latlon
        assert samplepreds.dim() == target.dim() + 1
        assert samplepreds.dim() in (3, 4)
        T, G, S = samplepreds.shape[0], samplepreds.shape[1], samplepreds.shape[-1]
        L = samplepreds.shape[2] if samplepreds.dim() == 4 else 1
        sp = _check(samplepreds, tuple(samplepreds.shape), "samplepreds")
        tg = _check(target, tuple(samplepreds.shape[:-1]), "target")
        nb = _lib.lib().csa_eval_scratch_bytes(T, G, L, S)
        scratch = torch.empty(nb, dtype=torch.uint8, device=sp.device)
        out = torch.empty((L,) if avg_grid else (G, L), device=sp.device)
        rc = _lib.lib().csa_eval_crps(T, G, L, S, _ptr(sp), _ptr(tg), int(bool(avg_grid)),
                                      ctypes.c_void_p(scratch.data_ptr()), _ptr(out), _stream(sp))
        if rc != 0:
            raise RuntimeError(f"csa_eval_crps failed ({rc}): {_lib.last_error()}")
        if samplepreds.dim() == 3:
            return out.reshape(()) if avg_grid else out.reshape(G)
        return out

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
