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


@pytest.mark.gpu
def test_hip_crps_gradient_matches_reference_autograd():
    """d CRPS / d (ensemble outputs): the native backward against autograd through the reference's own function (torch.cdist
    backward), via the same call a training loop makes: loss = CRPS(...); loss.backward()."""
    from climsim_amd.metrics import CRPS
    g = np.load(os.path.join(GOLDEN, "crps.npz"))
    for i in range(3):
        T, B, E, seed = (int(v) for v in g[f"case{i}.cfg"])
        y, ys, yp, yps = (t.cuda() for t in inputs(T, B, E, 60, 5, 8, seed))
        yp.requires_grad_(True); yps.requires_grad_(True)
        loss = 3.0 * CRPS(y, ys, yp, yps, T, beta=1, alpha=float(g[f"case{i}.alpha"]))
        loss.backward()
        for got, ref in ((yp.grad, g[f"case{i}.d_pred"]), (yps.grad, g[f"case{i}.d_sfc_pred"])):
            got = got.cpu().numpy() / 3.0
            assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), i
    # a member identical to the truth (zero distance) contributes a zero skill gradient, not NaN
    y, ys, yp, yps = (t.cuda() for t in inputs(1, 3, 2, 60, 5, 8, 5))
    yp = yp.clone(); yps = yps.clone()
    yp[0:3] = y; yps[0:3] = ys
    yp.requires_grad_(True); yps.requires_grad_(True)
    CRPS(y, ys, yp, yps, 1).backward()
    assert torch.isfinite(yp.grad).all() and torch.isfinite(yps.grad).all()


@pytest.mark.gpu
def test_ensemble_crps_training_step_through_the_stochastic_layer():
    """rnn/utils.py:1065-1075,1213: inputs replicated member-major, one noise draw per member, loss = CRPS of the ensemble,
    loss.backward(), optimiser step -- here on the stochastic LSTM layer alone (its native BPTT + the native CRPS gradient
    composed by autograd).  The composed gradient is checked by a directional finite difference, and a few SGD steps lower
    the score."""
    from climsim_amd.layers import MyStochasticLSTMLayer4
    from climsim_amd.metrics import CRPS
    g = torch.Generator().manual_seed(17)
    T, B, E, nx, H = 60, 4, 2, 64, 64
    w = (torch.rand(nx + H, 5 * H, generator=g) * 2 - 1) / H ** 0.5
    layer = MyStochasticLSTMLayer4(w, H, max_rows=T * B * E, requires_grad=True)
    x = torch.randn(T, B, nx, generator=g).cuda() * 0.5
    h0, c0 = torch.zeros(B * E, H).cuda(), torch.zeros(B * E, H).cuda()
    eps = torch.randn(T, E * B, H, generator=g).cuda()
    y = torch.randn(B, T, H, generator=g).cuda() * 0.3              # "truth": (B, nlev, ny) with ny = H here
    ys = torch.zeros(B, 1).cuda()

    def score():
        xe = torch.repeat_interleave(x.unsqueeze(1), E, dim=1).flatten(1, 2)      # (T, E*B, nx), member-major as upstream
        out, _ = layer(xe, (h0, c0), eps=eps)                                      # (T, E*B, H)
        return CRPS(y, ys, out.permute(1, 0, 2).contiguous(), torch.zeros(E * B, 1, device="cuda", requires_grad=True), 1)

    loss0 = score()
    loss0.backward()
    grad = layer.weight_encoder.grad.clone()
    assert torch.isfinite(grad).all() and float(grad.abs().max()) > 0
    # directional finite difference along the gradient (float32: a relative step of 1e-2 of the weight scale)
    d = grad / grad.norm()
    hstep = 2e-3
    with torch.no_grad():
        layer.weight_encoder.add_(hstep * d); layer.sync_params(); lp = float(score())
        layer.weight_encoder.add_(-2 * hstep * d); layer.sync_params(); lm = float(score())
        layer.weight_encoder.add_(hstep * d); layer.sync_params()
    fd = (lp - lm) / (2 * hstep)
    assert abs(fd - float(grad.norm())) <= 0.05 * float(grad.norm()), (fd, float(grad.norm()))
    # plain SGD on the score
    losses = [float(loss0)]
    for _ in range(5):
        layer.weight_encoder.grad = None
        l = score()
        l.backward()
        with torch.no_grad():
            layer.weight_encoder.add_(-0.5 * layer.weight_encoder.grad)
        layer.sync_params()
        losses.append(float(l))
    assert losses[-1] < losses[0], losses
