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
