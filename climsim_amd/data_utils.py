"""Device-side mirror of climsim_utils/data_utils.py (the loader API the north star keeps): same class name, constructor
signature, variable tables, `set_to_*_vars` selections and method names as the reference, float32 torch tensors on the GPU
where the reference has numpy arrays.

    constructor, tables            data_utils.py:47-566     (variable lists :178-400, var_lens :402-477, constants :158-170)
    set_to_v1 / v2 / v2_rh / vx / v4_rnn / v4 / v5_vars      :568-652
    derived inputs of get_xrdata   :654-707  -> derive_inputs() (csrc/derive.hip): state_rh, liq_partition, state_qn[..._prvphy]
    calc_MAE / RMSE / R2 / bias / CRPS   :1843-1935          (csrc/evalm.hip)
    reshape_input_for_cnn / reshape_target_for_cnn / reshape_target_from_cnn   :2104-2175  (csrc/cnn_api.hip)

Pinned by outputs of the reference module itself (tests/golden/make_golden_data_utils.py -> data_utils_api.json,
data_utils_golden.npz).  What stays host-side and outside this package: xarray / netCDF / HDF5 file reading and writing
(get_xrdata's open_dataset, save_as_h5 / save_as_npy, the tf.data generator), plotting and the pandas metric tables."""
import ctypes

import numpy as np
import torch

from . import _lib
from .emulator import _check, _ptr


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _values(x):
    """grid_info entries may be xarray DataArrays (.values), numpy arrays or sequences."""
    return np.asarray(getattr(x, "values", x))


# ---- variable tables, composed from their building blocks (E3SM variable names; data_utils.py:178-400) ---------------------------
_SFC_FLUX = "pbuf_SOLIN pbuf_LHFLX pbuf_SHFLX".split()
_SFC_V2 = ["state_ps"] + _SFC_FLUX + ("pbuf_TAUX pbuf_TAUY pbuf_COSZRS cam_in_ALDIF cam_in_ALDIR cam_in_ASDIF cam_in_ASDIR "
                                      "cam_in_LWUP cam_in_ICEFRAC cam_in_LANDFRAC cam_in_OCNFRAC cam_in_SNOWHICE cam_in_SNOWHLAND").split()
_GASES = "pbuf_ozone pbuf_CH4 pbuf_N2O".split()
_WIND = ["state_u", "state_v"]
_DYN = [p + v for p in ("", "tm_") for v in ("state_t_dyn", "state_q0_dyn", "state_u_dyn")]
_TM_SFC = "tm_state_ps tm_pbuf_SOLIN tm_pbuf_LHFLX tm_pbuf_SHFLX tm_pbuf_COSZRS clat slat icol".split()
_CAM_OUT = ["cam_out_" + v for v in "NETSW FLWDS PRECSC PRECC SOLS SOLL SOLSD SOLLD".split()]


def _prvphy(species):
    return [f"{p}state_{v}_prvphy" for p in ("", "tm_") for v in ["t"] + species + ["u"]]


V1_INPUTS = ["state_t", "state_q0001", "state_ps"] + _SFC_FLUX
V1_OUTPUTS = ["ptend_t", "ptend_q0001"] + _CAM_OUT
V2_INPUTS = ["state_t", "state_q0001", "state_q0002", "state_q0003"] + _WIND + _SFC_V2 + _GASES
V2_RH_INPUTS = ["state_t", "state_rh", "state_q0002", "state_q0003"] + _WIND + _GASES + _SFC_V2
V4_INPUTS = (["state_t", "state_rh", "state_q0002", "state_q0003"] + _WIND + _DYN + _prvphy(["q0001", "q0002", "q0003"])
             + _GASES + _SFC_V2 + _TM_SFC)
V4_RNN_INPUTS = [v for v in V4_INPUTS if "prvphy" not in v and "icol" not in v]
V5_INPUTS = (["state_t", "state_rh", "state_qn", "liq_partition"] + _WIND + _DYN + _prvphy(["q0001", "qn"]) + _GASES + _SFC_V2
             + _TM_SFC)
V2_OUTPUTS = ["ptend_t", "ptend_q0001", "ptend_q0002", "ptend_q0003", "ptend_u", "ptend_v"] + _CAM_OUT
V5_OUTPUTS = ["ptend_t", "ptend_q0001", "ptend_qn", "ptend_u", "ptend_v"] + _CAM_OUT
_PROFILE_VARS = (["state_t", "state_rh", "state_q0001", "state_q0002", "state_q0003", "state_qn", "liq_partition"] + _WIND
                 + [p + v for p in ("", "tm_") for v in ("state_t_dyn", "state_q0_dyn", "state_u_dyn")] + ["state_v_dyn"]
                 + [f"{p}state_{v}_prvphy" for p in ("", "tm_") for v in ("t", "q0001", "q0002", "q0003", "qn", "u")]
                 + _GASES + ["ptend_t", "ptend_q0001", "ptend_q0002", "ptend_q0003", "ptend_qn", "ptend_u", "ptend_v"])
_SCALAR_VARS = _SFC_V2 + _TM_SFC + _CAM_OUT + ["pbuf_SOLIN_pm", "pbuf_COSZRS_pm"]
# (selection name) -> (inputs, outputs, ps_index, input_feature_len, target_feature_len, full_vars, full_vars_v5); :568-652
_SETS = {"v1": (V1_INPUTS, V1_OUTPUTS, 120, 124, 128, False, None), "v2": (V2_INPUTS, V2_OUTPUTS, 360, 557, 368, True, None),
         "v2_rh": (V2_RH_INPUTS, V2_OUTPUTS, 360, 557, 368, True, None), "vx": (V4_RNN_INPUTS, V2_OUTPUTS, None, None, None, True, None),
         "v4_rnn": (V4_RNN_INPUTS, V2_OUTPUTS, None, None, None, True, None), "v4": (V4_INPUTS, V2_OUTPUTS, 1500, 1525, 368, True, None),
         "v5": (V5_INPUTS, V5_OUTPUTS, 1380, 1405, 308, False, True)}


class data_utils:
    def __init__(self, grid_info=None, input_mean=None, input_max=None, input_min=None, output_scale=None,
                 ml_backend="pytorch", normalize=True, input_abbrev="mli", output_abbrev="mlo", save_h5=False, save_npy=True,
                 *, num_latlon=None, num_levels=60):
        """Reference signature (data_utils.py:47-58).  grid_info: mapping with 'lev', 'ncol', 'area', 'lat', 'lon', 'hyam',
        'hybm' (xarray Dataset, or a dict of arrays); it may be omitted when only the scores / adapters are used, in which case
        `num_latlon` (and `num_levels`) give the grid size."""
        if ml_backend != "pytorch":
            raise ImportError("climsim_amd.data_utils is the PyTorch-ROCm mirror: ml_backend must be 'pytorch'")
        self.input_abbrev, self.output_abbrev = input_abbrev, output_abbrev
        self.data_path, self.save_h5, self.save_npy = None, save_h5, save_npy
        self.input_vars, self.target_vars = [], []
        self.input_feature_len = self.target_feature_len = None
        self.grid_info = grid_info
        self.level_name, self.sample_name = "lev", "sample"
        if grid_info is not None:
            self.num_levels = len(_values(grid_info["lev"]))
            self.num_latlon = len(_values(grid_info["ncol"]))
            area = _values(grid_info["area"]).astype(np.float64)
            self.area_wgt = area / area.mean()
            lat, lon = _values(grid_info["lat"]), _values(grid_info["lon"])
            self.lats, self.lats_indices = np.unique(lat, return_index=True)
            self.lons, self.lons_indices = np.unique(lon, return_index=True)
            self.indextolatlon = {i: (lat[i % self.num_latlon], lon[i % self.num_latlon]) for i in range(self.num_latlon)}
            self.hyam, self.hybm = _values(grid_info["hyam"]), _values(grid_info["hybm"])
        else:
            self.num_levels = int(num_levels)
            self.num_latlon = int(384 if num_latlon is None else num_latlon)
            self.area_wgt = self.hyam = self.hybm = None
        if num_latlon is not None and grid_info is not None and int(num_latlon) != self.num_latlon:
            raise ValueError("num_latlon disagrees with grid_info['ncol']")
        self.input_mean, self.input_max, self.input_min, self.output_scale = input_mean, input_max, input_min, output_scale
        self.normalize = normalize
        self.ml_backend, self.tf, self.torch, self.successful_backend_import = ml_backend, None, torch, True
        self.p0, self.ps_index = 1e5, None
        self.full_vars = self.full_vars_v5 = False
        # physical constants (E3SM shr_const_mod; data_utils.py:158-170)
        self.grav, self.cp, self.lv, self.lf = 9.80616, 1.00464e3, 2.501e6, 3.337e5
        self.lsub = self.lv + self.lf
        self.rho_air = 101325 / (6.02214e26 * 1.38065e-23 / 28.966) / 273.15
        self.rho_h20 = 1.e3
        self.v1_inputs, self.v1_outputs = list(V1_INPUTS), list(V1_OUTPUTS)
        self.v2_inputs, self.v2_rh_inputs, self.v2_outputs = list(V2_INPUTS), list(V2_RH_INPUTS), list(V2_OUTPUTS)
        self.v4_inputs, self.v4_rnn_inputs, self.v4_outputs = list(V4_INPUTS), list(V4_RNN_INPUTS), list(V2_OUTPUTS)
        self.v5_inputs, self.v5_outputs = list(V5_INPUTS), list(V5_OUTPUTS)
        self.var_lens = {**{v: self.num_levels for v in _PROFILE_VARS}, **{v: 1 for v in _SCALAR_VARS}}
        self.var_short_names = {"ptend_t": "$dT/dt$", "ptend_q0001": "$dq/dt$", **{v: v[len("cam_out_"):] for v in _CAM_OUT}}
        latent = self.lv * self.rho_h20
        self.target_energy_conv = {"ptend_t": self.cp, **{v: self.lv for v in ("ptend_q0001", "ptend_q0002", "ptend_q0003", "ptend_qn")},
                                   "ptend_wind": None, **{v: (latent if "PREC" in v else 1.) for v in _CAM_OUT}}
        self.metrics_dict = {"MAE": self.calc_MAE, "RMSE": self.calc_RMSE, "R2": self.calc_R2, "CRPS": self.calc_CRPS,
                             "bias": self.calc_bias}
        self.num_CRPS = 32

    # ---- variable-set selections (data_utils.py:568-652) ----
    def _select(self, name):
        inp, out, ps, nin, nout, full, full5 = _SETS[name]
        self.input_vars, self.target_vars = list(inp), list(out)
        if ps is not None:        # the vx / v4_rnn selections leave the three numbers untouched, as the reference does
            self.ps_index, self.input_feature_len, self.target_feature_len = ps, nin, nout
        self.full_vars = full
        if full5 is not None:
            self.full_vars_v5 = full5

    def set_to_v1_vars(self):
        self._select("v1")

    def set_to_v2_vars(self):
        self._select("v2")

    def set_to_v2_rh_vars(self):
        self._select("v2_rh")

    def set_to_vx_vars(self):
        self._select("vx")

    def set_to_v4_rnn_vars(self):
        self._select("v4_rnn")

    def set_to_v4_vars(self):
        self._select("v4")

    def set_to_v5_vars(self):
        self._select("v5")

    # ---- derived inputs (get_xrdata, data_utils.py:654-707) ----
    def derive_inputs(self, ds, file_vars=None):
        """`ds`: dict of float32 device tensors of one shape (what a file holds: state_t, state_q0001, state_pmid, ...).  Adds, in
        place, every variable of `file_vars` (default: self.input_vars) that the reference derives on read and `ds` lacks:
        state_rh, liq_partition, state_qn, state_qn_prvphy, tm_state_qn_prvphy.  Returns ds."""
        want = self.input_vars if file_vars is None else file_vars
        L = _lib.lib()

        def run(n, t=None, q1=None, pm=None, q2=None, q3=None, rh=None, liq=None, qn=None):
            rc = L.csa_derive_inputs(n, _ptr(t), _ptr(q1), _ptr(pm), _ptr(q2), _ptr(q3), _ptr(rh), _ptr(liq), _ptr(qn),
                                     _stream(next(x for x in (t, q2) if x is not None)))
            if rc != 0:
                raise RuntimeError(f"csa_derive_inputs failed ({rc}): {_lib.last_error()}")

        def get(k):
            t = ds[k]
            return _check(t, tuple(t.shape), k)
        need_rh = "state_rh" in want and "state_rh" not in ds
        need_liq = "liq_partition" in want and "liq_partition" not in ds
        if need_rh or need_liq:
            t = get("state_t")
            rh = torch.empty_like(t) if need_rh else None
            liq = torch.empty_like(t) if need_liq else None
            run(t.numel(), t=t, q1=get("state_q0001") if need_rh else None, pm=get("state_pmid") if need_rh else None, rh=rh, liq=liq)
            if need_rh:
                ds["state_rh"] = rh
            if need_liq:
                ds["liq_partition"] = liq
        for out, a, b in (("state_qn", "state_q0002", "state_q0003"), ("state_qn_prvphy", "state_q0002_prvphy", "state_q0003_prvphy"),
                          ("tm_state_qn_prvphy", "tm_state_q0002_prvphy", "tm_state_q0003_prvphy")):
            if out in want and out not in ds:
                q2, q3 = get(a), get(b)
                qn = torch.empty_like(q2)
                run(q2.numel(), q2=q2, q3=q3, qn=qn)
                ds[out] = qn
        return ds

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
