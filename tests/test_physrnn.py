"""physRNN "Hidden" model.  Oracle chain:
  shipped TorchScript artefact (run in the build container, outputs stored in tests/golden/physrnn_hidden.npz)
    -> CPU: the restatement oracle/physrnn_ref.py reproduces those outputs                 (pins the oracle)
    -> GPU: the HIP path (csa_phys_*, through the C-ABI) reproduces them too, and matches the float64 restatement at
            batch sizes the fixture does not hold.
Tolerance: 1e-5 x max|ref| per output block, or 3x the float32-vs-float64 error of the restatement itself where that is
larger (the decoder divides by pressure thickness and multiplies by 1200 s / scale factors: the artefact's own float32
result is ~1e-5 relative away from exact arithmetic on these synthetic inputs).  Measured (profiles/r2_physrnn_parity.txt): the
HIP error is 0.35-2.7 x that rounding level in every block of the three artefacts, i.e. the bound sits 1.1-9 x above what is
observed -- a regression of one order of magnitude fails."""
import os
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from make_golden_physrnn import inputs
from oracle import physrnn_ref

BLOCKS = [("out", c) for c in range(5)] + [("out_sfc", None), ("mem_out", None)]


def _load(name="physrnn_hidden"):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    P = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    return g, P


def _blocks(out, out_sfc, mem):
    d = {("out", c): out[..., c] for c in range(5)}
    d[("out_sfc", None)], d[("mem_out", None)] = out_sfc, mem
    return d


FIXTURES = [("physrnn_hidden", 2), ("physrnn_hidden_b", 1), ("physrnn_hidden_ep40", 1)]   # three checkpoints of the same graph


@pytest.mark.parametrize("fixture,ncase", FIXTURES)
def test_restatement_reproduces_the_artefact(fixture, ncase):
    g, P = _load(fixture)
    for i in range(ncase):
        B, seed = (int(v) for v in g[f"case{i}.cfg"])
        xm, xs, mem, xd = inputs(P, B, seed)
        got = _blocks(*physrnn_ref.forward(P, xm, xs, mem, xd, torch.from_numpy(g[f"case{i}.hx2"])))
        ref = _blocks(*(torch.from_numpy(g[f"case{i}.{k}"]) for k in ("out", "out_sfc", "mem_out")))
        for k in BLOCKS:
            err = (got[k] - ref[k]).abs().max().item()
            assert err <= 2e-5 * ref[k].abs().max().item(), (i, k, err, ref[k].abs().max().item())
        # structural facts of the decoder: nothing but radiative heating above the CRM top, stored water is one value per column
        assert torch.all(ref[("out", 1)][:, :10] == 0) and torch.all(ref[("out", 3)][:, :12] == 0)
        m = torch.from_numpy(g[f"case{i}.mem_out"])
        assert torch.all(m[:, :, -1] == m[:, :1, -1]) and torch.all(m[:, :, -1] >= 0)


def _hip_model(P, max_batch):
    from climsim_amd.physrnn import physical_RNN_autoreg
    return physical_RNN_autoreg(P, max_batch=max_batch)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture,ncase", FIXTURES)
def test_hip_physrnn_matches_the_artefact(fixture, ncase):
    g, P = _load(fixture)
    m = _hip_model(P, 64)
    for i in range(ncase):
        B, seed = (int(v) for v in g[f"case{i}.cfg"])
        xm, xs, mem, xd = inputs(P, B, seed)
        hx2 = torch.from_numpy(g[f"case{i}.hx2"])
        got = _blocks(*(t.cpu() for t in m([xm.cuda(), xs.cuda(), mem.cuda(), xd.cuda()], hx2=hx2.cuda())))
        ref = _blocks(*(torch.from_numpy(g[f"case{i}.{k}"]) for k in ("out", "out_sfc", "mem_out")))
        P64 = {k: v.double() for k, v in P.items()}
        r64 = _blocks(*physrnn_ref.forward(P64, xm.double(), xs.double(), mem.double(), xd.double(), hx2.double()))
        for k in BLOCKS:
            scale = ref[k].abs().max().item()
            noise = (ref[k].double() - r64[k]).abs().max().item()          # the artefact's own float32 rounding
            err = (got[k].double() - ref[k].double()).abs().max().item()
            assert err <= max(1e-5 * scale, 3 * noise), (i, k, err, scale, noise)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 2, 301, 384, 600])       # one / two columns per workgroup, and four on the matrix pipe (from 544)
def test_hip_physrnn_matches_restatement(B):
    g, P = _load()
    m = _hip_model(P, max(384, B))
    xm, xs, mem, xd = inputs(P, B, 50 + B)
    hx2 = torch.randn(B, 128, generator=torch.Generator().manual_seed(B))
    taps = {}
    P64 = {k: v.double() for k, v in P.items()}
    r64 = _blocks(*physrnn_ref.forward(P64, xm.double(), xs.double(), mem.double(), xd.double(), hx2.double(), taps=taps))
    taps32 = {}
    r32 = _blocks(*physrnn_ref.forward(P, xm, xs, mem, xd, hx2, taps=taps32))
    got = _blocks(*(t.cpu() for t in m([xm.cuda(), xs.cuda(), mem.cuda(), xd.cuda()], hx2=hx2.cuda())))
    # recurrent core first: level-major taps against the float64 restatement (|h| <= 1; 60 + 60 dependent steps)
    for which, name in ((1, "rnn1out"), (2, "rnn2out")):
        t = m.tap(which, B).cpu().permute(1, 0, 2).double()
        noise = (taps32[name].double() - taps[name]).abs().max().item()
        assert (t - taps[name]).abs().max().item() <= max(1e-5, 3 * noise), (name, noise)
    for k in BLOCKS:
        scale = r64[k].abs().max().item()
        noise = (r32[k].double() - r64[k]).abs().max().item()
        err = (got[k].double() - r64[k]).abs().max().item()
        assert err <= max(1e-5 * scale, 3 * noise), (B, k, err, scale, noise)
    assert all(torch.isfinite(v).all() for v in got.values())


@pytest.mark.gpu
def test_hip_physrnn_errors_and_rollout_state():
    g, P = _load()
    m = _hip_model(P, 16)
    xm, xs, mem, xd = (t.cuda() for t in inputs(P, 8, 3))
    with pytest.raises(RuntimeError):
        m([xm[:, :, :20], xs, mem, xd])
    with pytest.raises(RuntimeError):
        m([xm.cpu(), xs, mem, xd])
    with pytest.raises(RuntimeError):
        big = inputs(P, 17, 4)
        m([t.cuda() for t in big])
    # inputs are not modified; feeding the returned memory back works (autoregressive use)
    keep = [t.clone() for t in (xm, xs, mem, xd)]
    out, out_sfc, mem1 = m([xm, xs, mem, xd])
    assert all(torch.equal(a, b) for a, b in zip(keep, (xm, xs, mem, xd)))
    out2, _, mem2 = m([xm, xs, mem1, xd])
    assert torch.isfinite(out2).all() and torch.isfinite(mem2).all() and mem2.shape == mem.shape


@pytest.mark.gpu
@pytest.mark.parametrize("fixture,gen", [("physrnn_hidden", "inputs"), ("physrnn_rad", "inputs_rad")])
def test_hip_physrnn_postprocessing_matches_the_artefact(fixture, gen):
    """The module's exported `postprocessing(out, out_sfc, x_denorm)` (models.py:273-339) applied to the artefact's own outputs."""
    import make_golden_physrnn as G
    g, P = _load(fixture)
    m = _hip_model(P, 64)
    for i in range(2):
        B, seed = (int(v) for v in g[f"case{i}.cfg"])
        xd = getattr(G, gen)(P, B, seed)[3]
        out, out_sfc = torch.from_numpy(g[f"case{i}.out"]), torch.from_numpy(g[f"case{i}.out_sfc"])
        p6, psfc = m.postprocessing(out.cuda(), out_sfc.cuda(), xd.cuda())
        r6, rsfc = torch.from_numpy(g[f"case{i}.post_lev"]), torch.from_numpy(g[f"case{i}.post_sfc"])
        for c in range(6):
            scale = r6[..., c].abs().max().item()
            # columns 2, 3: (lf * (qn + 1200 dqn) - q) / 1200 -- a difference of two numbers that agree to ~1e-3: 1e-5 of the
            # LARGER of the tendency and q / 1200
            if c in (2, 3):
                scale = max(scale, xd[..., c].abs().max().item() / 1200.0)
            assert (p6[..., c].cpu() - r6[..., c]).abs().max().item() <= 1e-5 * scale, c
        assert (psfc.cpu() - rsfc).abs().max().item() <= 1e-6 * rsfc.abs().max().item()
