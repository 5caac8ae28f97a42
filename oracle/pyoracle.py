"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (climsim_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libclimsim_oracle.so")

_F = ctypes.POINTER(ctypes.c_float)

_INT_FIELDS = ["nlev", "nx", "nx_sfc", "ny", "ny_sfc", "nh1", "nh2", "nh_mem",
               "use_lstm", "legacy", "output_prune", "mp_mode", "snowhice_fix",
               "qinput_prune", "rh_prune", "scrub_inf", "scrub_out_nan", "q_input_mode", "v5_input"]
_CONST_FIELDS = ["xmean_lev", "xdiv_lev", "xmean_sca", "xdiv_sca", "lbd_qc", "lbd_qi", "lbd_qn",
                 "yscale_lev", "yscale_sca", "hyam", "hybm"]
_W_FIELDS = ["mlp_initial_w", "mlp_initial_b", "mlp_surface1_w", "mlp_surface1_b",
             "mlp_surface2_w", "mlp_surface2_b", "mlp_toa1_w", "mlp_toa1_b",
             "mlp_toa2_w", "mlp_toa2_b",
             "rnn1_w_ih", "rnn1_w_hh", "rnn1_b_ih", "rnn1_b_hh",
             "rnn2_w_ih", "rnn2_w_hh", "rnn2_b_ih", "rnn2_b_hh",
             "mlp_latent_w", "mlp_latent_b", "mlp_output_w", "mlp_output_b",
             "mlp_surface_output_w", "mlp_surface_output_b"]

# state_dict key -> struct field
_SD = {
    "mlp_initial.weight": "mlp_initial_w", "mlp_initial.bias": "mlp_initial_b",
    "mlp_surface1.weight": "mlp_surface1_w", "mlp_surface1.bias": "mlp_surface1_b",
    "mlp_surface2.weight": "mlp_surface2_w", "mlp_surface2.bias": "mlp_surface2_b",
    "mlp_toa1.weight": "mlp_toa1_w", "mlp_toa1.bias": "mlp_toa1_b",
    "mlp_toa2.weight": "mlp_toa2_w", "mlp_toa2.bias": "mlp_toa2_b",
    "rnn1.weight_ih_l0": "rnn1_w_ih", "rnn1.weight_hh_l0": "rnn1_w_hh",
    "rnn1.bias_ih_l0": "rnn1_b_ih", "rnn1.bias_hh_l0": "rnn1_b_hh",
    "rnn2.weight_ih_l0": "rnn2_w_ih", "rnn2.weight_hh_l0": "rnn2_w_hh",
    "rnn2.bias_ih_l0": "rnn2_b_ih", "rnn2.bias_hh_l0": "rnn2_b_hh",
    "mlp_latent.weight": "mlp_latent_w", "mlp_latent.bias": "mlp_latent_b",
    "mlp_output.weight": "mlp_output_w", "mlp_output.bias": "mlp_output_b",
    "mlp_surface_output.weight": "mlp_surface_output_w",
    "mlp_surface_output.bias": "mlp_surface_output_b",
}


class _CModel(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int) for n in _INT_FIELDS]
                + [(n, _F) for n in _CONST_FIELDS]
                + [(n, _F) for n in _W_FIELDS])


def build(force=False):
    if force or not os.path.exists(_LIB) or (
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "climsim_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        P = ctypes.POINTER(_CModel)
        _lib.oracle_model_forward.argtypes = [P, ctypes.c_int] + [_F] * 10
        _lib.oracle_preprocess.argtypes = [P, ctypes.c_int] + [_F] * 4
        _lib.oracle_wrapper_forward.argtypes = [P, ctypes.c_int] + [_F] * 6
        _lib.oracle_wrapper_forward_tuple.argtypes = [P, ctypes.c_int] + [_F] * 6
    return _lib


def _ptr(a):
    if a is None:
        return ctypes.cast(None, _F)
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_F)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class OracleModel:
    """Holds weights/constants (numpy fp32) + behaviour flags and calls the C oracle."""

    def __init__(self, consts, weights, *, legacy, use_lstm=True, nh_mem=None, mp_mode=1,
                 output_prune=False, snowhice_fix=False, qinput_prune=False, rh_prune=False,
                 scrub_inf=False, scrub_out_nan=False, q_input_mode=0, v5_input=False):
        self._keep = {}
        cm = _CModel()
        for k in _CONST_FIELDS:
            self._keep[k] = _c(consts.get(k))
            setattr(cm, k, _ptr(self._keep[k]))
        if v5_input and self._keep["lbd_qn"] is None:
            raise ValueError("v5_input needs the lbd_qn constants")
        for sd, f in _SD.items():
            a = _c(weights.get(sd))
            self._keep[f] = a
            setattr(cm, f, _ptr(a))
        w = self._keep
        G = 4 if use_lstm else 3
        cm.nlev = consts["xmean_lev"].shape[0]
        cm.nx = consts["xmean_lev"].shape[1]
        cm.nx_sfc = consts["xmean_sca"].shape[0]
        cm.ny = w["mlp_output_w"].shape[0]
        cm.ny_sfc = w["mlp_surface_output_w"].shape[0]
        cm.nh1 = w["rnn1_w_hh"].shape[1]
        cm.nh2 = w["rnn2_w_hh"].shape[1]
        assert w["rnn1_w_hh"].shape[0] == G * cm.nh1
        cm.nh_mem = (w["mlp_latent_w"].shape[0] if w["mlp_latent_w"] is not None else 0) \
            if nh_mem is None else nh_mem
        assert w["rnn1_w_ih"].shape[1] == cm.nh1 + cm.nh_mem
        cm.use_lstm = int(use_lstm)
        cm.legacy = int(legacy)
        cm.output_prune = int(output_prune)
        cm.mp_mode = int(mp_mode)
        cm.snowhice_fix = int(snowhice_fix)
        cm.qinput_prune = int(qinput_prune)
        cm.rh_prune = int(rh_prune)
        cm.scrub_inf = int(scrub_inf)
        cm.scrub_out_nan = int(scrub_out_nan)
        cm.q_input_mode = int(q_input_mode)
        cm.v5_input = int(v5_input)
        self.cm = cm

    @classmethod
    def from_npz(cls, path, **kw):
        d = np.load(path)
        consts = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
        weights = {k[2:]: d[k] for k in d.files if k.startswith("w.")}
        return cls(consts, weights, **kw)

    # -- calls -----------------------------------------------------------
    def n_out(self):
        return 6 * self.cm.nlev + self.cm.ny_sfc + self.cm.nlev * self.cm.nh_mem

    def wrapper_forward(self, x_main, x_sfc, mem_in=None, hx2=None, cx2=None):
        B = x_main.shape[0]
        x_main, x_sfc, mem_in, hx2, cx2 = map(_c, (x_main, x_sfc, mem_in, hx2, cx2))
        y = np.empty((B, self.n_out()), np.float32)
        rc = lib().oracle_wrapper_forward(ctypes.byref(self.cm), B, _ptr(x_main), _ptr(x_sfc),
                                          _ptr(mem_in), _ptr(hx2), _ptr(cx2), _ptr(y))
        if rc != 0:
            raise RuntimeError(f"oracle_wrapper_forward rc={rc}")
        return y

    def wrapper_forward_tuple(self, x_main, x_sfc, mem_in):
        B = x_main.shape[0]
        cm = self.cm
        x_main, x_sfc, mem_in = map(_c, (x_main, x_sfc, mem_in))
        nyo = 6 if cm.mp_mode != 0 else cm.ny
        out_lev = np.empty((B, cm.nlev, nyo), np.float32)
        out_sfc = np.empty((B, cm.ny_sfc), np.float32)
        mem_out = np.empty((cm.nlev, B, cm.nh_mem), np.float32)
        rc = lib().oracle_wrapper_forward_tuple(ctypes.byref(cm), B, _ptr(x_main), _ptr(x_sfc),
                                                _ptr(mem_in), _ptr(out_lev), _ptr(out_sfc), _ptr(mem_out))
        if rc != 0:
            raise RuntimeError(f"oracle_wrapper_forward_tuple rc={rc}")
        return out_lev, out_sfc, mem_out

    def preprocess(self, x_main, x_sfc):
        B = x_main.shape[0]
        x_main, x_sfc = _c(x_main), _c(x_sfc)
        xn = np.empty((B, self.cm.nlev, self.cm.nx), np.float32)
        xs = np.empty_like(x_sfc)
        lib().oracle_preprocess(ctypes.byref(self.cm), B, _ptr(x_main), _ptr(x_sfc), _ptr(xn), _ptr(xs))
        return xn, xs

    def model_forward(self, x_main_n, x_sfc_n, mem_in=None, hx2=None, cx2=None, taps=False):
        B = x_main_n.shape[0]
        cm = self.cm
        x_main_n, x_sfc_n, mem_in, hx2, cx2 = map(_c, (x_main_n, x_sfc_n, mem_in, hx2, cx2))
        out = np.empty((B, cm.nlev, cm.ny), np.float32)
        out_sfc = np.empty((B, cm.ny_sfc), np.float32)
        mem_out = None
        if cm.nh_mem > 0:
            shp = (B, cm.nlev, cm.nh_mem) if cm.legacy else (cm.nlev, B, cm.nh_mem)
            mem_out = np.empty(shp, np.float32)
        r1 = np.empty((B, cm.nlev, cm.nh1), np.float32) if taps else None
        r2 = np.empty((B, cm.nlev, cm.nh2), np.float32) if taps else None
        rc = lib().oracle_model_forward(ctypes.byref(cm), B, _ptr(x_main_n), _ptr(x_sfc_n), _ptr(mem_in),
                                        _ptr(hx2), _ptr(cx2), _ptr(out), _ptr(out_sfc), _ptr(mem_out),
                                        _ptr(r1), _ptr(r2))
        if rc != 0:
            raise RuntimeError(f"oracle_model_forward rc={rc}")
        return (out, out_sfc, mem_out, r1, r2) if taps else (out, out_sfc, mem_out)
