"""Device-side mirror of the reference's ensemble scores (rnn/metrics.py: CRPS :535-626, compute_spread_skill_ratio
:509-533, CRPS_l1 :628-699).  CRPS is differentiable w.r.t. the ensemble outputs (native backward, csa_crps_backward), so it
can be the training loss of the stochastic model as in rnn/utils.py:1213."""
import ctypes

import torch

from . import _lib
from .emulator import _check, _ptr


class _CRPSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred, y_sfc_pred, y, y_sfc, timesteps, beta, alpha):
        ctx.save_for_backward(y, y_sfc, y_pred, y_sfc_pred)
        ctx.cfg = (timesteps, beta, alpha)
        return _crps_value(y, y_sfc, y_pred, y_sfc_pred, timesteps, beta, alpha)[0]

    @staticmethod
    def backward(ctx, g):
        y, y_sfc, y_pred, y_sfc_pred = ctx.saved_tensors
        T, beta, alpha = ctx.cfg
        ns, L, F = y.shape
        B = ns // T
        E = y_pred.shape[0] // (T * B)
        d_p, d_s = torch.empty_like(y_pred), torch.empty_like(y_sfc_pred)
        rc = _lib.lib().csa_crps_backward(T, B, E, L * F, y_sfc.shape[-1], _ptr(y), _ptr(y_sfc), _ptr(y_pred), _ptr(y_sfc_pred),
                                          float(beta), float(alpha), float(g.item()), _ptr(d_p), _ptr(d_s),
                                          ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"csa_crps_backward failed ({rc}): {_lib.last_error()}")
        return d_p, d_s, None, None, None, None, None


def CRPS(y, y_sfc, y_pred, y_sfc_pred, timesteps, beta=1, alpha=1.0, return_terms=False):
    """Same arguments as the reference: y (T*B, nlev, ny), y_sfc (T*B, ny_sfc), y_pred (T*E*B, nlev, ny) ordered
    (time, member, column), y_sfc_pred (T*E*B, ny_sfc).  Returns the score (a 0-d tensor), optionally with its two terms.
    With y_pred / y_sfc_pred requiring grad the score is part of the autograd graph."""
    if not return_terms and torch.is_grad_enabled() and (y_pred.requires_grad or y_sfc_pred.requires_grad):
        ns, L, F = y.shape
        E = y_pred.shape[0] // ns
        return _CRPSFn.apply(_check(y_pred, (ns * E, L, F), "y_pred"), _check(y_sfc_pred, (ns * E, y_sfc.shape[-1]), "y_sfc_pred"),
                             _check(y, (ns, L, F), "y"), _check(y_sfc, (ns, y_sfc.shape[-1]), "y_sfc"), int(timesteps), beta, alpha)
    out = _crps_value(y, y_sfc, y_pred, y_sfc_pred, timesteps, beta, alpha)
    return (out[0], out[1], out[2]) if return_terms else out[0]


def _crps_value(y, y_sfc, y_pred, y_sfc_pred, timesteps, beta, alpha):
    """Same arguments as the reference: y (T*B, nlev, ny), y_sfc (T*B, ny_sfc), y_pred (T*E*B, nlev, ny) ordered
    (time, member, column), y_sfc_pred (T*E*B, ny_sfc).  Returns the score (a 0-d tensor), optionally with its two terms."""
    ns, L, F = y.shape
    B = ns // timesteps
    E = y_pred.shape[0] // (timesteps * B)
    y = _check(y, (ns, L, F), "y")
    y_sfc = _check(y_sfc, (ns, y_sfc.shape[-1]), "y_sfc")
    y_pred = _check(y_pred, (timesteps * E * B, L, F), "y_pred")
    y_sfc_pred = _check(y_sfc_pred, (timesteps * E * B, y_sfc.shape[-1]), "y_sfc_pred")
    scratch = torch.empty(2 * ns, device=y.device)
    out = torch.empty(3, device=y.device)
    rc = _lib.lib().csa_crps(timesteps, B, E, L * F, y_sfc.shape[-1], _ptr(y), _ptr(y_sfc), _ptr(y_pred), _ptr(y_sfc_pred),
                             float(beta), float(alpha), _ptr(scratch), _ptr(out),
                             ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"csa_crps failed ({rc}): {_lib.last_error()}")
    return out


def _ens_scores(y, y_sfc, y_pred, y_sfc_pred, timesteps):
    ns, L, F = y.shape
    B = ns // timesteps
    E = y_pred.shape[0] // (timesteps * B)
    y = _check(y, (ns, L, F), "y")
    y_sfc = _check(y_sfc, (ns, y_sfc.shape[-1]), "y_sfc")
    y_pred = _check(y_pred, (timesteps * E * B, L, F), "y_pred")
    y_sfc_pred = _check(y_sfc_pred, (timesteps * E * B, y_sfc.shape[-1]), "y_sfc_pred")
    scratch = torch.empty(4 * 1024, dtype=torch.float64, device=y.device)
    out = torch.empty(4, device=y.device)
    rc = _lib.lib().csa_spread_skill(timesteps, B, E, L * F, y_sfc.shape[-1], _ptr(y), _ptr(y_sfc), _ptr(y_pred),
                                     _ptr(y_sfc_pred), ctypes.c_void_p(scratch.data_ptr()), _ptr(out),
                                     ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"csa_spread_skill failed ({rc}): {_lib.last_error()}")
    return out


def compute_spread_skill_ratio(y, y_sfc, y_pred, y_sfc_pred, timesteps):
    """rnn/metrics.py:509-533: (spread, rmse) of the ensemble, same arguments and return order."""
    out = _ens_scores(y, y_sfc, y_pred, y_sfc_pred, timesteps)
    return out[0], out[1]


def CRPS_l1(y, y_sfc, y_pred, y_sfc_pred, timesteps, beta=1):
    """rnn/metrics.py:628-699: skill - 0.5 * |member 0 - member 1| (the reference assumes two members; beta is unused there too)."""
    return _ens_scores(y, y_sfc, y_pred, y_sfc_pred, timesteps)[2]
