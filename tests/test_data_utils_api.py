"""climsim_amd.data_utils against the reference's own climsim_utils/data_utils.py (tests/golden/make_golden_data_utils.py):
constructor tables and variable-set selections (CPU: pure host logic), derived inputs, evaluation scores and CNN adapters
(GPU, through the C ABI)."""
import inspect
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from synth import CRPS_CASES, EVAL_CASES, checksum, crps_inputs, derived_inputs, eval_inputs

API = json.load(open(os.path.join(GOLDEN, "data_utils_api.json")))


def _grid(ncol=384, nlev=60):
    g = np.load(os.path.join(GOLDEN, "v4_memory_model.npz"))
    gd = np.load(os.path.join(GOLDEN, "data_utils_golden.npz"))
    return {"lev": np.arange(nlev), "ncol": np.arange(ncol), "area": gd["grid_area"],
            "lat": np.repeat(np.linspace(-80, 80, 24), 16), "lon": np.tile(np.linspace(0, 337.5, 16), 24),
            "hyam": g["c.hyam"], "hybm": g["c.hybm"]}


def test_constructor_signature_and_tables_match_the_reference():
    from climsim_amd.data_utils import data_utils
    names = list(inspect.signature(data_utils.__init__).parameters)[1:12]
    assert names == ["grid_info", "input_mean", "input_max", "input_min", "output_scale", "ml_backend", "normalize",
                     "input_abbrev", "output_abbrev", "save_h5", "save_npy"]           # data_utils.py:47-58
    du = data_utils(_grid(), None, None, None, None, ml_backend="pytorch")
    assert du.num_levels == API["num_levels"] and du.num_latlon == API["num_latlon"] and du.p0 == API["p0"]
    for k, v in API["constants"].items():
        assert getattr(du, k) == v, k
    assert du.var_lens == API["var_lens"]
    assert du.var_short_names == API["var_short_names"]
    assert du.target_energy_conv == API["target_energy_conv"]
    assert du.num_CRPS == API["num_CRPS"] and set(du.metrics_dict) == {"MAE", "RMSE", "R2", "CRPS", "bias"}
    gd = np.load(os.path.join(GOLDEN, "data_utils_golden.npz"))
    assert np.allclose(du.area_wgt, gd["area_wgt"], rtol=1e-15)
    with pytest.raises(ImportError):
        data_utils(_grid(), None, None, None, None, ml_backend="tensorflow")      # as upstream where TensorFlow is absent


@pytest.mark.parametrize("name", sorted(API["sets"]))
def test_variable_set_selections_match_the_reference(name):
    from climsim_amd.data_utils import data_utils
    du = data_utils(_grid(), None, None, None, None, ml_backend="pytorch")
    getattr(du, f"set_to_{name}_vars")()
    ref = API["sets"][name]
    for k in ("input_vars", "target_vars", "ps_index", "input_feature_len", "target_feature_len", "full_vars", "full_vars_v5"):
        assert getattr(du, k) == ref[k], (name, k)
    if ref["input_feature_len"] is not None:        # the flat vector length follows from the tables
        assert sum(du.var_lens[v] for v in du.input_vars) == ref["input_feature_len"]
        assert sum(du.var_lens[v] for v in du.target_vars) == ref["target_feature_len"]
        if name != "v2_rh":      # upstream keeps v2's ps_index for the re-ordered v2_rh list (state_ps sits at 540 there)
            assert sum(du.var_lens[v] for v in du.input_vars[:du.input_vars.index("state_ps")]) == ref["ps_index"]


@pytest.mark.gpu
def test_derived_inputs_match_the_reference_formulas():
    from climsim_amd.data_utils import data_utils
    gd = np.load(os.path.join(GOLDEN, "data_utils_golden.npz"))
    g = _grid()
    tair, pmid, q1, q2, q3 = derived_inputs(g["hyam"], g["hybm"])
    assert checksum(tair, pmid, q1, q2, q3) == gd["der.checksum"]
    du = data_utils(g, None, None, None, None, ml_backend="pytorch")
    du.set_to_v5_vars()
    d = lambda a: torch.from_numpy(a).cuda()
    ds = {"state_t": d(tair), "state_pmid": d(pmid), "state_q0001": d(q1), "state_q0002": d(q2), "state_q0003": d(q3),
          "state_q0002_prvphy": d(q3), "state_q0003_prvphy": d(q2), "tm_state_q0002_prvphy": d(q2), "tm_state_q0003_prvphy": d(q2)}
    du.derive_inputs(ds)
    rh = ds["state_rh"].cpu().numpy().astype(np.float64)
    # missing cells (NaN, +inf, -inf temperatures at [3,7], [4,8], [5,9]) propagate exactly as through the reference's numpy lines
    ok = np.isfinite(gd["der.state_rh"])
    assert (~ok).sum() == 3 and np.array_equal(np.isnan(rh), ~ok)
    assert np.abs(rh[ok] / gd["der.state_rh"][ok] - 1).max() <= 2e-7    # float64 polynomials, one float32 rounding at the end
    assert np.array_equal(ds["liq_partition"].cpu().numpy(), gd["der.liq_partition"], equal_nan=True)
    assert np.array_equal(ds["state_qn"].cpu().numpy(), gd["der.state_qn"])
    assert np.array_equal(ds["state_qn_prvphy"].cpu().numpy(), q3 + q2)
    assert np.array_equal(ds["tm_state_qn_prvphy"].cpu().numpy(), q2 + q2)
    # saturation pressures alone: T -> rh with q = qvs-scale 1 isolates eliq / eice at the branch points too
    T = gd["sat.T"]
    om = np.clip((T - np.float32(253.16)) / np.float32(20.0), 0, 1).astype(np.float32)
    esat = om.astype(np.float64) * gd["sat.eliq"] + (1 - om.astype(np.float64)) * gd["sat.eice"]
    one = np.ones_like(T)
    ds2 = du.derive_inputs({"state_t": d(T), "state_q0001": d(one), "state_pmid": d(one)}, ["state_rh"])
    want = 1.0 / ((287 * esat) / 461.0)
    assert np.abs(ds2["state_rh"].cpu().numpy() / want - 1).max() <= 2e-7
    # present variables are left alone; missing sources raise
    keep = ds["state_rh"]
    assert du.derive_inputs(ds)["state_rh"] is keep
    with pytest.raises(KeyError):
        du.derive_inputs({"state_t": d(T)}, ["state_rh"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(EVAL_CASES))
@pytest.mark.parametrize("avg_grid", [True, False])
def test_scores_match_the_reference_methods(tag, avg_grid):
    from climsim_amd.data_utils import data_utils
    gd = np.load(os.path.join(GOLDEN, "data_utils_golden.npz"))
    pred, target = eval_inputs(tag)
    assert checksum(pred, target) == gd[f"m{tag}.checksum"]
    du = data_utils(num_latlon=384)
    got = du.calc_all(torch.from_numpy(pred).cuda(), torch.from_numpy(target).cuda(), avg_grid)
    mae_scale = np.abs(gd[f"m{tag}.MAE.{int(avg_grid)}"]).max()
    for k in ("MAE", "RMSE", "R2", "bias"):
        ref = gd[f"m{tag}.{k}.{int(avg_grid)}"]
        g = got[k].cpu().numpy().astype(np.float64)
        assert g.shape == ref.shape, (k, g.shape, ref.shape)
        scale = max(np.abs(ref).max(), mae_scale if k == "bias" else 0.0)
        assert np.abs(g - ref).max() <= 2e-6 * scale, (k, np.abs(g - ref).max(), scale)      # float64 sums, fp32 result


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(CRPS_CASES))
@pytest.mark.parametrize("avg_grid", [True, False])
def test_crps_matches_the_reference_method(tag, avg_grid):
    from climsim_amd.data_utils import data_utils
    gd = np.load(os.path.join(GOLDEN, "data_utils_golden.npz"))
    sp, target = crps_inputs(tag)
    assert checksum(sp, target) == gd[f"c{tag}.checksum"]
    du = data_utils(num_latlon=384)
    got = du.calc_CRPS(torch.from_numpy(sp).cuda(), torch.from_numpy(target).cuda(), avg_grid).cpu().numpy()
    ref = gd[f"c{tag}.CRPS.{int(avg_grid)}"]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.gpu
def test_cnn_adapters_match_the_reference_static_methods():
    from climsim_amd.data_utils import data_utils
    gd = np.load(os.path.join(GOLDEN, "data_utils_golden.npz"))
    d = lambda k: torch.from_numpy(gd[k]).cuda()
    assert np.array_equal(data_utils.reshape_input_for_cnn(d("cnn.x")).cpu().numpy(), gd["cnn.x_cnn"])
    assert np.array_equal(data_utils.reshape_target_for_cnn(d("cnn.y")).cpu().numpy(), gd["cnn.y_cnn"])
    back = data_utils.reshape_target_from_cnn(d("cnn.pred_cnn")).cpu().numpy()
    assert np.array_equal(back[:, :120], gd["cnn.pred_flat"][:, :120])
    assert np.abs(back[:, 120:] - gd["cnn.pred_flat"][:, 120:]).max() <= 2e-7     # level mean: summation order
