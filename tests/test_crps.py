"""Ensemble score: HIP kernel vs scalars produced by the reference's own rnn/metrics.py::CRPS."""
import os
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from make_golden_crps import inputs


def _torch_crps(y, ys, yp, yps, T, beta, alpha):     # restatement of rnn/metrics.py:568-608 (test infrastructure)
    ns, L, F = y.shape
    B = ns // T
    E = yp.shape[0] // (T * B)
    z = torch.cat((yp.reshape(T, E, B, L * F).transpose(1, 2).reshape(T * B, E, L * F),
                   yps.reshape(T, E, B, -1).transpose(1, 2).reshape(T * B, E, -1)), -1)
    zt = torch.cat((y.reshape(T * B, 1, L * F), ys.reshape(T * B, 1, -1)), -1)
    eps = (1 - alpha) / E
    mse = torch.cdist(zt, z).mean() / z.size(-1) ** 0.5
    var = ((1 - eps) * torch.cdist(z, z).mean(0).sum()) / (E * (E - 1)) / z.size(-1) ** 0.5
    return beta * 2 * mse - var


def test_restatement_matches_reference_scalars():
    g = np.load(os.path.join(GOLDEN, "crps.npz"))
    for i in range(3):
        T, B, E, seed = (int(v) for v in g[f"case{i}.cfg"])
        y, ys, yp, yps = inputs(T, B, E, 60, 5, 8, seed)
        v = _torch_crps(y.double(), ys.double(), yp.double(), yps.double(), T, 1, float(g[f"case{i}.alpha"]))
        assert abs(v.item() - float(g[f"case{i}.crps"])) <= 2e-6 * abs(float(g[f"case{i}.crps"]))


@pytest.mark.gpu
def test_hip_crps_matches_reference():
    from climsim_amd.metrics import CRPS
    g = np.load(os.path.join(GOLDEN, "crps.npz"))
    for i in range(3):
        T, B, E, seed = (int(v) for v in g[f"case{i}.cfg"])
        y, ys, yp, yps = inputs(T, B, E, 60, 5, 8, seed)
        v = CRPS(y.cuda(), ys.cuda(), yp.cuda(), yps.cuda(), T, beta=1, alpha=float(g[f"case{i}.alpha"]))
        assert abs(v.item() - float(g[f"case{i}.crps"])) <= 1e-5 * abs(float(g[f"case{i}.crps"])), i


def test_spread_skill_restatement_matches_reference_scalars():
    # CPU: float64 restatement of rnn/metrics.py:509-533 and :628-699 against the scalars the reference itself produced
    g = np.load(os.path.join(GOLDEN, "crps.npz"))
    for i in range(3):
        T, B, E, seed = (int(v) for v in g[f"case{i}.cfg"])
        y, ys, yp, yps = (t.double() for t in inputs(T, B, E, 60, 5, 8, seed))
        z = torch.cat((yp.reshape(T, E, B, -1), yps.reshape(T, E, B, -1)), -1)
        zt = torch.cat((y.reshape(T, B, -1), ys.reshape(T, B, -1)), -1)
        spread = z.var(dim=1).mean().sqrt() * ((E + 1) / E) ** 0.5
        rmse = (z.mean(dim=1) - zt).square().mean().sqrt()
        l1 = (z - zt[:, None]).abs().mean() - 0.5 * (z[:, 0] - z[:, 1]).abs().mean()
        for name, v in (("spread", spread), ("rmse", rmse), ("crps_l1", l1)):
            assert abs(v.item() - float(g[f"case{i}.{name}"])) <= 2e-6 * abs(float(g[f"case{i}.{name}"])), (i, name)


@pytest.mark.gpu
def test_hip_spread_skill_and_crps_l1_match_reference():
    from climsim_amd.metrics import compute_spread_skill_ratio, CRPS_l1
    g = np.load(os.path.join(GOLDEN, "crps.npz"))
    for i in range(3):
        T, B, E, seed = (int(v) for v in g[f"case{i}.cfg"])
        y, ys, yp, yps = (t.cuda() for t in inputs(T, B, E, 60, 5, 8, seed))
        sp, rm = compute_spread_skill_ratio(y, ys, yp, yps, T)
        l1 = CRPS_l1(y, ys, yp, yps, T)
        for name, v in (("spread", sp), ("rmse", rm), ("crps_l1", l1)):
            ref = float(g[f"case{i}.{name}"])
            assert abs(v.item() - ref) <= 1e-5 * abs(ref), (i, name, v.item(), ref)


@pytest.mark.gpu
def test_hip_spread_skill_large_and_single_member():
    from climsim_amd.metrics import compute_spread_skill_ratio
    T, B, E = 3, 384, 4
    y, ys, yp, yps = inputs(T, B, E, 60, 5, 8, 7)
    z = torch.cat((yp.double().reshape(T, E, B, -1), yps.double().reshape(T, E, B, -1)), -1)
    zt = torch.cat((y.double().reshape(T, B, -1), ys.double().reshape(T, B, -1)), -1)
    spread = (z.var(dim=1).mean().sqrt() * ((E + 1) / E) ** 0.5).item()
    rmse = (z.mean(dim=1) - zt).square().mean().sqrt().item()
    sp, rm = compute_spread_skill_ratio(y.cuda(), ys.cuda(), yp.cuda(), yps.cuda(), T)
    assert abs(sp.item() - spread) <= 1e-5 * spread and abs(rm.item() - rmse) <= 1e-5 * rmse
    # one member: zero spread (the reference's var of one sample is nan; the kernel reports 0), rmse of that member
    y, ys, yp, yps = inputs(2, 9, 1, 60, 5, 8, 8)
    sp, rm = compute_spread_skill_ratio(y.cuda(), ys.cuda(), yp.cuda(), yps.cuda(), 2)
    ref = torch.cat(((yp - y).reshape(18, -1), (yps - ys).reshape(18, -1)), -1).double().square().mean().sqrt().item()
    assert sp.item() == 0.0 and abs(rm.item() - ref) <= 1e-5 * ref
