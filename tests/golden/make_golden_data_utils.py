#!/usr/bin/env python3
"""Golden data from the reference's own climsim_utils/data_utils.py, run in the build container.

Import notes.  data_utils.py imports xarray, netCDF4 and h5py at module top for its file I/O (get_xrdata, save_as_h5, ...).
None is installed here and none is reached by what is pinned below: the constructor's variable tables, the set_to_*_vars
selections, eliq / eice (numpy polynomials), calc_MAE / RMSE / R2 / bias / CRPS (numpy reductions) and the three CNN reshape
adapters (numpy stack / mean).  The three names are bound to inert placeholder modules before the import (any call into them
raises); the constructor gets a small duck-typed grid table (arrays with .values / .mean(dim=...), exactly what it reads).
What CANNOT be called is the RH / liq_partition / qn derivation inside get_xrdata (:662-697), which opens a netCDF file: its
six arithmetic lines are restated here around the reference's own eliq / eice (marked below).
Outputs: data_utils_api.json (tables) and data_utils_golden.npz (arrays) -- data only."""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)


class _Inert(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def refuse(*a, **k):
            raise RuntimeError(f"{self.__name__}.{name}: placeholder, not available in this container")
        return refuse


class _Arr:
    """The slice of the xarray.DataArray interface data_utils.__init__ touches."""

    def __init__(self, a):
        self.values = np.asarray(a)

    def __len__(self):
        return len(self.values)

    def mean(self, dim=None):
        return _Arr(self.values.mean())

    def __truediv__(self, o):
        return _Arr(self.values / (o.values if isinstance(o, _Arr) else o))


def main():
    for n in ("xarray", "netCDF4", "h5py"):
        sys.modules.setdefault(n, _Inert(n))
    sys.path.insert(0, REF)
    from climsim_utils import data_utils as ref_mod
    g = np.load(os.path.join(OUT, "v4_memory_model.npz"))
    r = np.random.default_rng(12)
    ncol, nlev = 384, 60
    grid = {"lev": _Arr(np.arange(nlev)), "ncol": _Arr(np.arange(ncol)), "area": _Arr(r.uniform(0.5, 1.5, ncol)),
            "lat": _Arr(np.repeat(np.linspace(-80, 80, 24), 16)), "lon": _Arr(np.tile(np.linspace(0, 337.5, 16), 24)),
            "hyam": _Arr(g["c.hyam"]), "hybm": _Arr(g["c.hybm"])}
    du = ref_mod.data_utils(grid, None, None, None, None, ml_backend="pytorch")
    api = {"num_levels": du.num_levels, "num_latlon": du.num_latlon, "p0": du.p0,
           "constants": {k: getattr(du, k) for k in ("grav", "cp", "lv", "lf", "lsub", "rho_air", "rho_h20")},
           "var_lens": du.var_lens, "var_short_names": du.var_short_names,
           "target_energy_conv": du.target_energy_conv, "num_CRPS": du.num_CRPS, "sets": {}}
    for name in ("v1", "v2", "v2_rh", "vx", "v4_rnn", "v4", "v5"):
        d = ref_mod.data_utils(grid, None, None, None, None, ml_backend="pytorch")
        getattr(d, f"set_to_{name}_vars")()
        api["sets"][name] = {"input_vars": list(d.input_vars), "target_vars": list(d.target_vars), "ps_index": d.ps_index,
                             "input_feature_len": d.input_feature_len, "target_feature_len": d.target_feature_len,
                             "full_vars": d.full_vars, "full_vars_v5": d.full_vars_v5}
    json.dump(api, open(os.path.join(OUT, "data_utils_api.json"), "w"), indent=1, sort_keys=True)

    arrs = {"area_wgt": du.area_wgt.astype(np.float64), "grid_area": grid["area"].values}
    # ---- evaluation scores on float32 data, as float64 numpy evaluates them (the arrays the reference scores are float64
    # after reweighting; float32 storage keeps the fixture small) ------------------------------------------------------------
    from synth import CRPS_CASES, EVAL_CASES, checksum, crps_inputs, derived_inputs, eval_inputs
    for tag in EVAL_CASES:          # inputs are regenerated from their seeds by the tests (checksums stored)
        pred, target = eval_inputs(tag)
        arrs[f"m{tag}.checksum"] = checksum(pred, target)
        p64, t64 = pred.astype(np.float64), target.astype(np.float64)
        for ag in (True, False):
            for k, f in (("MAE", du.calc_MAE), ("RMSE", du.calc_RMSE), ("R2", du.calc_R2), ("bias", du.calc_bias)):
                arrs[f"m{tag}.{k}.{int(ag)}"] = np.asarray(f(p64, t64, avg_grid=ag))
    for tag in CRPS_CASES:
        sp, target = crps_inputs(tag)
        arrs[f"c{tag}.checksum"] = checksum(sp, target)
        for ag in (True, False):
            arrs[f"c{tag}.CRPS.{int(ag)}"] = np.asarray(du.calc_CRPS(sp.astype(np.float64), target.astype(np.float64), avg_grid=ag))
    # ---- CNN adapters ----------------------------------------------------------------------------------------------------------
    x = r.standard_normal((9, 124)).astype(np.float32)
    y = r.standard_normal((9, 128)).astype(np.float32)
    yc = r.standard_normal((9, 60, 10)).astype(np.float32)
    arrs.update({"cnn.x": x, "cnn.x_cnn": ref_mod.data_utils.reshape_input_for_cnn(x), "cnn.y": y,
                 "cnn.y_cnn": ref_mod.data_utils.reshape_target_for_cnn(y), "cnn.pred_cnn": yc,
                 "cnn.pred_flat": ref_mod.data_utils.reshape_target_from_cnn(yc)})
    # ---- saturation pressures (the module's own functions) and the derived inputs of get_xrdata ---------------------------------
    T = np.concatenate([np.linspace(150.0, 330.0, 721), [185.0, 273.15, 273.16, 253.16]]).astype(np.float32)
    arrs["sat.T"], arrs["sat.eliq"], arrs["sat.eice"] = T, ref_mod.eliq(T), ref_mod.eice(T)
    tair, pmid, q1, q2, q3 = derived_inputs(g["c.hyam"], g["c.hybm"])
    # data_utils.py:662-673, 684-697 restated around the reference's eliq / eice (get_xrdata itself needs a netCDF file)
    omega = np.maximum(0, np.minimum(1, (tair - 253.16) / (273.16 - 253.16)))
    esat = omega * ref_mod.eliq(tair) + (1 - omega) * ref_mod.eice(tair)
    qvs = (287 * esat) / (461 * pmid)
    arrs.update({"der.checksum": checksum(tair, pmid, q1, q2, q3),
                 "der.state_rh": q1 / qvs, "der.liq_partition": omega, "der.state_qn": q2 + q3})
    np.savez_compressed(os.path.join(OUT, "data_utils_golden.npz"), **arrs)
    print({k: (v.shape, str(v.dtype)) for k, v in arrs.items() if k.startswith(("sat", "der"))})
    print("sets:", {k: (len(v["input_vars"]), len(v["target_vars"])) for k, v in api["sets"].items()})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present: golden fixtures can only be regenerated in the build container")
    main()
