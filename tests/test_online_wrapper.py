"""Generic online wrapper forward(x (B, n_in)) -> (B, 368) (SURVEY section 8b "generic online", 8f #4):
HIP path (csa_online_*, through the C-ABI) against the torch restatement oracle/online_ref.py.
Parity is UNPINNED by reference outputs (modulus is not installed, no trained weights ship): random nn.Linear-style
weights, synthetic normalisation vectors of the reference's shapes.  Tolerance: 1e-5 x max|ref| per output block."""
import numpy as np
import pytest
import torch

from oracle import online_ref


def _setup(n_in, seed=0):
    g = torch.Generator().manual_seed(seed)
    sub = torch.randn(n_in, generator=g) * 0.1
    div = torch.rand(n_in, generator=g) + 0.5
    osc = torch.rand(368, generator=g) * 3 + 0.5
    lbd_qc = torch.rand(60, generator=g) * 5e4 + 1e3
    lbd_qi = torch.rand(60, generator=g) * 5e4 + 1e3
    return sub, div, osc, lbd_qc, lbd_qi


def _input(B, n_in, seed):
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(B, n_in, generator=g)
    x[:, 60:120] = torch.rand(B, 60, generator=g) * 1.5 - 0.1          # RH incl. values outside [0, 1.2]
    x[:, 120:240] = torch.rand(B, 120, generator=g) * 1e-4              # cloud liquid / ice, kg/kg
    return x


def test_oracle_wrapper_properties():
    # CPU: the restatement zeroes exactly the ranges of the notebook and is insensitive to the pruned inputs
    n_in = 557
    sub, div, osc, lqc, lqi = _setup(n_in)
    from climsim_amd.online import MLP
    m = MLP(n_in, 368, [384, 1024, 640], 3, output_prune=True, strato_lev_out=12)
    ws, bs = m.weights()
    ws, bs = [torch.from_numpy(w) for w in ws], [torch.from_numpy(b) for b in bs]
    x = _input(5, n_in, 0)
    y = online_ref.new_model(x, ws, bs, sub, div, osc, lqc, lqi)
    for a, b in ((60, 75), (120, 148), (180, 195), (240, 255), (300, 315)):
        assert torch.all(y[:, a:b] == 0)
    assert torch.all(y[:, -8:] >= 0)
    x2 = x.clone()
    x2[:, 120:135] = 7.0
    x2[:, 180:195] = 3.0
    assert torch.equal(y, online_ref.new_model(x2, ws, bs, sub, div, osc, lqc, lqi))
    x3 = x.clone()
    x3[0, 3] = float("nan")
    x3[1, 300] = float("inf")
    assert torch.isfinite(online_ref.new_model(x3, ws, bs, sub, div, osc, lqc, lqi)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n_in,hidden,B", [(557, [384, 1024, 640], 384), (557, [384, 1024, 640], 1), (1525, [512, 512], 1500),
                                           (557, [384, 1024, 640], 2700)])
def test_online_wrapper_parity(n_in, hidden, B):
    from climsim_amd.online import MLP, NewModel
    sub, div, osc, lqc, lqi = _setup(n_in, seed=n_in)
    m = MLP(n_in, 368, hidden, len(hidden), output_prune=True, strato_lev_out=12)
    ws, bs = m.weights()
    wt, bt = [torch.from_numpy(w).double() for w in ws], [torch.from_numpy(b).double() for b in bs]
    x = _input(B, n_in, B)
    x[0, 5] = float("nan")
    x[B - 1, 400] = float("-inf")
    ref = online_ref.new_model(x.double(), wt, bt, sub.double(), div.double(), osc.double(), lqc.double(), lqi.double())
    ref32 = online_ref.new_model(x, [w.float() for w in wt], [b.float() for b in bt], sub, div, osc, lqc, lqi)
    nm = NewModel(m, sub.numpy(), div.numpy(), osc.numpy(), lqc.numpy(), lqi.numpy(), max_batch=max(B, 8))
    xd = x.cuda()
    x_before = xd.clone()
    y = nm(xd).cpu()
    assert torch.equal(torch.nan_to_num(xd, 0.0, 1.0, -1.0), torch.nan_to_num(x_before, 0.0, 1.0, -1.0)), "input was modified"
    for a, b in ((0, 60), (60, 120), (120, 180), (180, 240), (240, 300), (300, 360), (360, 368)):
        r = ref[:, a:b]
        err = (y[:, a:b].double() - r).abs().max().item()
        e32 = (ref32[:, a:b].double() - r).abs().max().item()
        tol = 1e-5 * r.abs().max().item()
        assert err <= max(tol, 2 * e32), (a, b, err, tol, e32)
    for a, b in ((60, 75), (120, 148), (180, 195), (240, 255), (300, 315)):
        assert torch.all(y[:, a:b] == 0)


@pytest.mark.gpu
def test_online_wrapper_errors():
    from climsim_amd.online import MLP, NewModel
    sub, div, osc, lqc, lqi = _setup(557)
    m = MLP(557, 368, [64, 64], 2, output_prune=False)
    nm = NewModel(m, sub.numpy(), div.numpy(), osc.numpy(), lqc.numpy(), lqi.numpy(), max_batch=16)
    with pytest.raises(RuntimeError):
        nm(torch.zeros(4, 556, device="cuda"))
    with pytest.raises(RuntimeError):
        nm(torch.zeros(17, 557, device="cuda"))
    with pytest.raises(RuntimeError):
        nm(torch.zeros(4, 557))
