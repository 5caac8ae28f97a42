"""physRNN "Hidden" radiation graphs (second slice: `use_physrad`).  Six shipped artefacts, four graphs:
  physrnn_rad          num4050_BEST          mp_ncol 4, MCICA sub-column sampling
  physrnn_rad_nomcica  num71535_BEST         mp_ncol 16, g-point g = sub-column g
  physrnn_rad_liqfrac  num83000_ep20         + learned cloud liquid-fraction head
  physrnn_rad_stoch_*  num5730_BEST, num62104_BEST, num62104_BEST_ep11    num4050's graph + stochastic third RNN
and the first geometry of the physRNN_physRad-* family (97 of the 114 shipped models):
  physrad16_nh96       physRNN_physRad-16_nreg16_*neur96-96_xv4_mp1_num20600_BEST: the same graph with GRU 96/96 and no third RNN
  physrad16_nh112_cld  physRNN_physRad-16_nreg16_*neur112-112_xv4_mp1_num88955_BEST: as nh112_a/b, cloud LW optical depth from a learned
                       Linear(19, 16) + ReLU instead of the liquid / ice rule
  physrad4_a/b         physRNN_physRad-16_nreg4_*neur112-112_xv4_mp1_num35741_BEST (21 inputs), _num95220_BEST (16 inputs): four regions, the
                       g-points sample the three cloudy ones (MCICA), otherwise as the nh112 graphs
  physrad16_nh112_a/b  physRNN_physRad-16_nreg16_*neur112-112_xv4_mp1_num34341_BEST, _num37201_BEST: GRU 112/112, 16 level inputs, no
                       third RNN, a later revision of the serialised scheme (the LW downward sweep has its own source)
  physrad16_a / b / c  physRNN_physRad-16_nreg16_*neur128-128_xv4_mp1_num14751_BEST, _num55617_BEST, _num55617_ep12: 16 regions,
                       region 0 clear sky, no sub-grid temperature, liquid-fraction head, stochastic third RNN, q/(1-q) mixing
                       ratio, rnn_mem level-major
Oracle chain:
  shipped TorchScript artefact, run in the build container (torch.jit.load, CPU; its internal randn draws reproduced by
  re-seeding), outputs stored in tests/golden/<fixture>.npz
    -> CPU: the restatement oracle/physrnn_rad_ref.py reproduces those outputs                  (pins the oracle)
    -> GPU: the HIP path (csa_phys_rad_create + csa_phys_forward, through the C-ABI) reproduces them too, and matches the
            float64 restatement at batch sizes the fixture does not hold.
Tolerance, per output block: max(1e-5 x max|ref|, 6 x noise), noise = the float32 rounding level of the block, measured as the
distance of the artefact (and of the float32 restatement) from the float64 restatement -- see _noise_level; HIP is held to
that against the float64 restatement, and to 7 x against the artefact itself (triangle inequality).  Measured:
profiles/r2_physrnn_parity.txt -- the HIP error is 0.5-4.6 x the level in every block of every artefact.  The radiation scheme raises MLP outputs to the 8th power, multiplies by ~1e22 molecules/cm2 and
differences net fluxes over thin layers, so the artefact itself sits 3e-5..5e-5 (relative to the block maximum) away from
exact arithmetic on these inputs (the stochastic graphs, whose third RNN scales its noise by exp(z/2), up to 2.5e-4); the
float64 restatement is separately required to be within 3e-3 of the artefact so that "noise" cannot hide a wrong formula
(the bound is set by cells near the two-stream singularity k * mu0 = 1, physics_rad.py:139, where the direct-beam terms are a
small difference of large numbers divided by 1 - (k mu0)^2: the reference guards the denominator at 1e-7 and clamps the result,
but up to there rounding is amplified; seen in these fixtures: 2.3e-3 on one column's SOLL).  Formula identity, free of
rounding, was checked when the fixtures were generated: the artefacts run in float64 (module.double(), default dtype
float64) agree with the float64 restatement to 1e-14 on every output of the graphs without such a cell and to 4e-7 otherwise
(tests/golden/make_golden_physrnn.py::check_float64)."""
import os
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from make_golden_physrnn import inputs_rad
from oracle import physrnn_rad_ref

BLOCKS = [("out", c) for c in range(5)] + [("out_sfc", c) for c in range(8)] + [("mem_out", None)]


def _load(name="physrnn_rad"):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    P = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    return g, P


def _blocks(out, out_sfc, mem):
    d = {("out", c): out[..., c] for c in range(5)}
    d.update({("out_sfc", c): out_sfc[:, c] for c in range(8)})
    d[("mem_out", None)] = mem
    return d


def _ref64(P, xm, xs, mem, xd, hx2, taps=None, **noise):
    P64 = {k: v.double() for k, v in P.items()}
    return _blocks(*physrnn_rad_ref.forward(P64, xm.double(), xs.double(), mem.double(), xd.double(), hx2.double(), taps=taps,
                                            **{k: v.double() for k, v in noise.items()}))


NOISE_FACTOR = 6


def _noise_level(*realisations_then_exact):
    """Rounding level of a block: the larger max|x - exact| of the float32 realisations at hand (the artefact's and / or the float32
    restatement's) against the float64 restatement.  One realisation's maximum over a block is itself a noisy estimate of the level
    -- two of them differ by 2-4x in these fixtures, most where a column has a cell near the two-stream singularity -- hence two
    realisations where available and NOISE_FACTOR = 6; measured HIP error / level: 0.5-4.6 (profiles/r2_physrnn_parity.txt)."""
    *xs, exact = realisations_then_exact
    return max((x.double() - exact).abs().max().item() for x in xs)


FIXTURES = [("physrnn_rad", 2), ("physrnn_rad_nomcica", 1), ("physrnn_rad_liqfrac", 1), ("physrnn_rad_stoch_a", 1),
            ("physrnn_rad_stoch_b", 1), ("physrnn_rad_stoch_c", 1), ("physrad16_a", 1), ("physrad16_b", 1), ("physrad16_c", 1),
            ("physrad16_nh96", 1), ("physrad16_nh112_a", 1), ("physrad16_nh112_b", 1),
            ("physrad4_a", 1), ("physrad4_b", 1), ("physrad16_nh112_cld", 1), ("physrad16_e3sm", 1), ("physrad16_e3sm_cld", 1), ("physrad16_b_gpu", 1)]


def _noise(g, i):
    """The artefact's N(0,1) draws; for the physRad fixture also its rnn3 output `srnn` (teacher forcing): that artefact's
    stochastic layer is chaotic on these synthetic inputs -- a 1e-6 difference in its input grows to 0.1 over the 50 levels
    (float32 vs float64 of the restatement itself) -- so the end-to-end outputs of two float32 implementations do not agree;
    the layer is checked step by step (test_*_rnn3_steps) and everything after it with its output supplied."""
    return {k: torch.from_numpy(g[f"case{i}.{k}"]) for k in ("hx1", "eps3", "srnn") if f"case{i}.{k}" in g.files}


def _rnn3_step_inputs(P, r2, hx1, srnn, eps3):
    """Every step of rnn3 as an independent batch row: input level t, previous state = the reference's output of level t-1."""
    T, B, H = srnn.shape
    x = r2.transpose(0, 1).reshape(1, T * B, H)
    h0 = torch.cat([hx1.unsqueeze(0), srnn[:-1]], 0).reshape(T * B, H)
    return x, h0, eps3.reshape(1, T * B, H)


@pytest.mark.parametrize("fixture", ["physrad16_a", "physrad16_b", "physrad16_c", "physrad16_b_gpu"])
def test_restatement_rnn3_steps_reproduce_the_artefact(fixture):
    g, P = _load(fixture)
    B, seed = (int(v) for v in g["case0.cfg"])
    xm, xs, mem, xd = inputs_rad(P, B, seed)
    nz, taps = _noise(g, 0), {}
    physrnn_rad_ref.forward(P, xm, xs, mem, xd, torch.from_numpy(g["case0.hx2"]), taps=taps, **nz)
    x, h0, eps = _rnn3_step_inputs(P, taps["rnn2out"], nz["hx1"], nz["srnn"], nz["eps3"])
    got = physrnn_rad_ref.stochastic_gru(x, h0, eps, P["rnn3.weight_ih"], P["rnn3.weight_zh"], P["rnn3.weight_encoder"])
    err = (got.reshape(nz["srnn"].shape) - nz["srnn"]).abs().max().item()
    assert err <= 2e-5 * nz["srnn"].abs().max().item(), err
    # and the reason for the teacher forcing: free-running, float32 and float64 of the same restatement part ways
    free32 = physrnn_rad_ref.stochastic_gru(taps["rnn2out"].transpose(0, 1), nz["hx1"], nz["eps3"], P["rnn3.weight_ih"], P["rnn3.weight_zh"],
                                            P["rnn3.weight_encoder"])
    free64 = physrnn_rad_ref.stochastic_gru(*(t.double() for t in (taps["rnn2out"].transpose(0, 1), nz["hx1"], nz["eps3"], P["rnn3.weight_ih"],
                                                                  P["rnn3.weight_zh"], P["rnn3.weight_encoder"])))
    d = (free32.double() - free64).abs().amax((1, 2))
    assert d[0] < 1e-4 and (fixture != "physrad16_a" or d[-1] > 100 * d[0])


def _draw_noise(P, B, seed):
    if "rnn3.weight_ih" not in P:
        return {}
    gen = torch.Generator().manual_seed(seed)
    nh = P["rnn2.weight_hh_l0"].shape[1]
    return {"hx1": torch.randn(B, nh, generator=gen), "eps3": torch.randn(50, B, nh, generator=gen)}


@pytest.mark.parametrize("fixture,ncase", FIXTURES)
def test_restatement_reproduces_the_artefact(fixture, ncase):
    g, P = _load(fixture)
    for i in range(ncase):
        B, seed = (int(v) for v in g[f"case{i}.cfg"])
        xm, xs, mem, xd = inputs_rad(P, B, seed)
        hx2, nz = torch.from_numpy(g[f"case{i}.hx2"]), _noise(g, i)
        got = _blocks(*physrnn_rad_ref.forward(P, xm, xs, mem, xd, hx2, **nz))
        r64 = _ref64(P, xm, xs, mem, xd, hx2, **nz)
        ref = _blocks(*(torch.from_numpy(g[f"case{i}.{k}"]) for k in ("out", "out_sfc", "mem_out")))
        for k in BLOCKS:
            scale = ref[k].abs().max().item()
            noise = (ref[k].double() - r64[k]).abs().max().item()
            err = (got[k] - ref[k]).abs().max().item()
            own = (got[k].double() - r64[k]).abs().max().item()        # the float32 restatement's own rounding
            assert noise <= 3e-3 * scale, (i, k, noise, scale)
            assert err <= max(1e-5 * scale, 3 * (noise + own)), (i, k, err, scale, noise, own)
        # structure: night columns have no shortwave at the surface; heating reaches the levels above the CRM top;
        # the moisture tendencies do not
        night = (xs[:, 6] * P["xdiv_sca"][6] + P["xmean_sca"][6]) < 1e-6
        assert night.any() and (~night).any()
        for c in (0, 4, 5, 6, 7):
            assert torch.all(ref[("out_sfc", c)][night] == 0) and torch.all(ref[("out_sfc", c)][~night] >= 0)
            assert torch.any(ref[("out_sfc", c)][~night] > 0)
        assert (ref[("out", 0)][:, :10] != 0).float().mean() > 0.9 and torch.all(ref[("out", 1)][:, :10] == 0)


@pytest.mark.parametrize("fixture", [f for f, _ in FIXTURES])
def test_restatement_float64_formula_identity(fixture):
    """Formula identity free of rounding, as a STANDING guard of the loosest parity gate in the tree (round-2 review, item 10): the
    artefact itself, converted to float64 and run with float64 default dtype in the build container
    (tests/golden/make_golden_physrnn.py::check_float64 stored its outputs in physrnn_f64.npz), against the float64 restatement on
    the same inputs and the same re-seeded draws.  Measured at generation time: 1e-14 on graphs without a cell near the two-stream
    singularity k mu0 = 1 and without the third RNN, up to 5e-7 otherwise (the reference's 1e-7 guard of 1 - (k mu0)^2 leaves a
    1e5-fold amplification of the last float64 digits there; the stochastic layer multiplies its noise by exp(z / 2)) -- a wrong
    formula shows up at 1e-3 or worse.  Asserted: 2e-6 of every block's maximum."""
    f64 = np.load(os.path.join(GOLDEN, "physrnn_f64.npz"))
    g, P = _load(fixture)
    P64 = {k: v.double() for k, v in P.items()}
    xm, xs, mem, xd = (t.double() for t in inputs_rad(P, 8, 77))
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(5)
        nh = P["rnn2.weight_hh_l0"].shape[1]
        hx2 = torch.randn(8, nh)
        kw = dict(hx1=torch.randn(8, nh), eps3=torch.randn(50, 8, nh)) if "rnn3.weight_ih" in P else {}
        got = physrnn_rad_ref.forward(P64, xm, xs, mem, xd, hx2, **kw)
    finally:
        torch.set_default_dtype(torch.float32)
    for k, a in zip(("out", "out_sfc", "mem_out"), got):
        b = torch.from_numpy(f64[f"{fixture}.{k}"])
        assert a.dtype == torch.float64 and a.shape == b.shape
        assert ((a - b).abs().max() / b.abs().max()).item() <= 2e-6, k


def test_subcolumn_sampling_properties():
    """physics_rad.py:533: every sub-column gets floor or ceil of p*G g-points, G in total, in sub-column order."""
    gen = torch.Generator().manual_seed(5)
    p = torch.softmax(3.0 * torch.randn(500, 4, generator=gen), 1)
    p[0] = torch.tensor([0.25, 0.25, 0.25, 0.25])             # ties
    p[1] = torch.tensor([1.0, 0.0, 0.0, 0.0])
    sub = physrnn_rad_ref.subcolumn_of_gpoint(p, 16)
    assert sub.shape == (500, 16) and sub.min() >= 0 and sub.max() <= 3
    assert torch.all(sub[:, 1:] >= sub[:, :-1])
    counts = torch.stack([(sub == j).sum(1) for j in range(4)], 1).float()
    assert torch.all((counts - p * 16).abs() < 1.0 + 1e-5) and torch.all(counts.sum(1) == 16)
    assert torch.equal(counts[0], torch.full((4,), 4.0)) and torch.equal(counts[1], torch.tensor([16.0, 0, 0, 0]))


def _hip_model(P, max_batch):
    from climsim_amd.physrnn import physical_RNN_autoreg
    m = physical_RNN_autoreg(P, max_batch=max_batch)
    assert m.use_physrad
    return m


def _cuda(d):
    return {k: v.cuda() for k, v in d.items()}


def _run(m, xm, xs, mem, xd, **kw):
    """Fixtures and the restatement keep rnn_mem as (B, 50, 16); the physRad graphs take and return it level-major."""
    lm = getattr(m, "physrad", False)
    mem_in = mem.transpose(0, 1).contiguous() if lm else mem
    if "srnn" in kw:
        kw = {"hx2": kw["hx2"], "_srnn": kw["srnn"]}
    out, out_sfc, mem_out = m([xm.cuda(), xs.cuda(), mem_in.cuda(), xd.cuda()], **_cuda(kw))
    return out.cpu(), out_sfc.cpu(), (mem_out.transpose(0, 1) if lm else mem_out).cpu()


@pytest.mark.gpu
@pytest.mark.parametrize("fixture,ncase", FIXTURES)
def test_hip_radiation_graph_matches_the_artefact(fixture, ncase):
    g, P = _load(fixture)
    m = _hip_model(P, 64)
    for i in range(ncase):
        B, seed = (int(v) for v in g[f"case{i}.cfg"])
        xm, xs, mem, xd = inputs_rad(P, B, seed)
        hx2, nz = torch.from_numpy(g[f"case{i}.hx2"]), _noise(g, i)
        got = _blocks(*_run(m, xm, xs, mem, xd, hx2=hx2, **nz))
        ref = _blocks(*(torch.from_numpy(g[f"case{i}.{k}"]) for k in ("out", "out_sfc", "mem_out")))
        r64 = _ref64(P, xm, xs, mem, xd, hx2, **nz)
        r32 = _blocks(*physrnn_rad_ref.forward(P, xm, xs, mem, xd, hx2, **nz))
        for k in BLOCKS:
            scale = ref[k].abs().max().item()
            noise = _noise_level(ref[k], r32[k], r64[k])
            # HIP is a third float32 realisation: no further from exact arithmetic than NOISE_FACTOR x the rounding of the other
            # two, hence (triangle inequality) within NOISE_FACTOR + 1 of the artefact
            e64, eref = (got[k].double() - r64[k]).abs().max().item(), (got[k].double() - ref[k].double()).abs().max().item()
            assert e64 <= max(1e-5 * scale, NOISE_FACTOR * noise), (i, k, e64, scale, noise)
            assert eref <= max(1e-5 * scale, (NOISE_FACTOR + 1) * noise), (i, k, eref, scale, noise)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture,B", [("physrnn_rad", 1), ("physrnn_rad", 2), ("physrnn_rad", 301), ("physrnn_rad", 384),
                                       ("physrnn_rad_nomcica", 301), ("physrnn_rad_liqfrac", 384), ("physrnn_rad_stoch_a", 2),
                                       ("physrnn_rad_stoch_b", 301), ("physrnn_rad_stoch_c", 384), ("physrad16_a", 2),
                                       ("physrad16_a", 301), ("physrad16_a", 384), ("physrad16_nh96", 384),
                                       ("physrad16_nh112_a", 301), ("physrad16_nh112_b", 384), ("physrad4_a", 384), ("physrad4_b", 301),
                                       ("physrad16_nh112_cld", 384), ("physrad16_e3sm", 384),
                                       # (num88741 is not in this list: on these synthetic inputs a 1e-7 relative input perturbation moves its SW surface
                                       #  fluxes by 1e-2 in several columns of a 301-column batch -- no float32 bound holds; its parity is the B = 8 fixture)
                                       # from 544 columns the GRUs run on the matrix-pipe four-column kernel (GRU 128 / 96 / 112)
                                       ("physrnn_rad", 600), ("physrad16_e3sm", 600), ("physrad16_nh96", 600), ("physrad16_nh112_a", 600)])
def test_hip_radiation_graph_matches_restatement(fixture, B):
    g, P = _load(fixture)
    m = _hip_model(P, max(384, B))
    xm, xs, mem, xd = inputs_rad(P, B, 70 + B)
    hx2 = torch.randn(B, P["rnn2.weight_hh_l0"].shape[1], generator=torch.Generator().manual_seed(B))
    nz = _draw_noise(P, B, 900 + B)
    taps = {}
    r64 = _ref64(P, xm, xs, mem, xd, hx2, taps, **nz)
    if fixture.startswith("physrad") and "rnn3.weight_ih" in P:            # chaotic rnn3 (see _noise): everything after it with the float64 restatement's output
        nz = {"srnn": taps["srnn"].float()}
        r64 = _ref64(P, xm, xs, mem, xd, hx2, taps, **nz)
    taps32 = {}
    r32 = _blocks(*physrnn_rad_ref.forward(P, xm, xs, mem, xd, hx2, taps=taps32, **nz))
    got = _blocks(*_run(m, xm, xs, mem, xd, hx2=hx2, **nz))
    # recurrent core first: rnn2 output over the 50 CRM levels against the float64 restatement (|h| <= 1; 50 + 50 dependent steps)
    t2 = m.tap(2, B).cpu().permute(1, 0, 2).double()
    noise = (taps32["rnn2out"].double() - taps["rnn2out"]).abs().max().item()
    assert (t2 - taps["rnn2out"]).abs().max().item() <= max(1e-5, 3 * noise), noise
    for k in BLOCKS:
        scale = r64[k].abs().max().item()
        noise = (r32[k].double() - r64[k]).abs().max().item()
        err = (got[k].double() - r64[k]).abs().max().item()
        assert err <= max(1e-5 * scale, NOISE_FACTOR * noise), (B, k, err, scale, noise)
    assert all(torch.isfinite(v).all() for v in got.values())


@pytest.mark.gpu
def test_hip_radiation_graph_errors_and_rollout_state():
    g, P = _load()
    m = _hip_model(P, 16)
    xm, xs, mem, xd = (t.cuda() for t in inputs_rad(P, 8, 3))
    with pytest.raises(RuntimeError):
        m([xm[:, :, :20], xs, mem, xd])
    with pytest.raises(RuntimeError):
        m([xm, xs[:, :14], mem, xd])
    with pytest.raises(RuntimeError):
        m([t.cuda() for t in inputs_rad(P, 17, 4)])
    keep = [t.clone() for t in (xm, xs, mem, xd)]
    hx2 = torch.randn(8, 128, device="cuda")
    out, out_sfc, mem1 = m([xm, xs, mem, xd], hx2=hx2)
    assert all(torch.equal(a, b) for a, b in zip(keep, (xm, xs, mem, xd)))
    out_b, out_sfc_b, mem1_b = m([xm, xs, mem, xd], hx2=hx2)         # deterministic: same inputs, same bits
    assert torch.equal(out, out_b) and torch.equal(out_sfc, out_sfc_b) and torch.equal(mem1, mem1_b)
    out2, _, mem2 = m([xm, xs, mem1, xd])
    assert torch.isfinite(out2).all() and torch.isfinite(mem2).all() and mem2.shape == mem.shape
    # a wrong-geometry state_dict is refused, not reinterpreted
    bad = dict(P)
    bad["mlp_qv_crm.weight"] = torch.zeros(8, 128)
    from climsim_amd.physrnn import physical_RNN_autoreg
    with pytest.raises(RuntimeError):
        physical_RNN_autoreg(bad, max_batch=8)


@pytest.mark.gpu
def test_hip_sw_gas_optics_flags_are_validated():
    """csa_phys_rad_create: the SW gas-optics block belongs to the 16-region physRad graphs (with the liquid-fraction head, or with both
    learned cloud-optics layers); a state_dict that mixes the families is refused, not reinterpreted."""
    from climsim_amd.physrnn import physical_RNN_autoreg
    g, P = _load("physrad16_e3sm")
    bad = {k: v for k, v in P.items() if not k.startswith("mlp_liq_frac_crm")}                  # Slingo / EC optics need the learned liquid fraction
    with pytest.raises(RuntimeError):
        physical_RNN_autoreg(bad, max_batch=8)
    g2, P2 = _load("physrad16_e3sm_cld")
    bad = {k: v for k, v in P2.items() if not k.startswith("cloud_optics_lw")}                  # learned SW cloud optics without the LW layer
    with pytest.raises(RuntimeError):
        physical_RNN_autoreg(bad, max_batch=8)
    bad = dict(P)
    bad["gas_optics_model_sw1.mlp1.weight"] = torch.zeros(16, 7)                               # not the shipped 7 -> 32 -> 32 -> 112 model
    with pytest.raises(RuntimeError):
        physical_RNN_autoreg(bad, max_batch=8)
    m = _hip_model(P, 8)                                                                        # the untouched state_dict still builds
    assert m.physrad


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["physrad16_a", "physrad16_b", "physrad16_c", "physrad16_b_gpu"])
def test_hip_rnn3_steps_reproduce_the_artefact(fixture):
    g, P = _load(fixture)
    m = _hip_model(P, 16)
    B, seed = (int(v) for v in g["case0.cfg"])
    xm, xs, mem, xd = inputs_rad(P, B, seed)
    nz, taps = _noise(g, 0), {}
    physrnn_rad_ref.forward(P, xm, xs, mem, xd, torch.from_numpy(g["case0.hx2"]), taps=taps, **nz)
    x, h0, eps = _rnn3_step_inputs(P, taps["rnn2out"], nz["hx1"], nz["srnn"], nz["eps3"])
    got = m.debug_rnn3(x.cuda(), h0.cuda(), eps.cuda()).cpu().reshape(nz["srnn"].shape)
    err = (got - nz["srnn"]).abs().max().item()
    assert err <= 2e-5 * nz["srnn"].abs().max().item(), err
    # free-running, the first levels still agree with the artefact (before the divergence has grown)
    free = m.debug_rnn3(taps["rnn2out"].transpose(0, 1).contiguous().cuda(), nz["hx1"].cuda(), nz["eps3"].cuda()).cpu()
    assert (free[:5] - nz["srnn"][:5]).abs().max().item() <= 5e-4


@pytest.mark.gpu
def test_hip_stochastic_graph_noise_handling():
    """add_stochastic_layer graph: explicit draws make a call reproducible, absent draws are made per call (as the reference's
    forward does), a wrong noise shape is refused."""
    g, P = _load("physrnn_rad_stoch_a")
    m = _hip_model(P, 16)
    assert m.stochastic
    xm, xs, mem, xd = (t.cuda() for t in inputs_rad(P, 8, 5))
    nz = _cuda(_draw_noise(P, 8, 1))
    hx2 = torch.randn(8, 128, device="cuda")
    a = m([xm, xs, mem, xd], hx2=hx2, **nz)
    b = m([xm, xs, mem, xd], hx2=hx2, **nz)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    c = m([xm, xs, mem, xd], hx2=hx2)
    assert not torch.equal(a[0], c[0]) and torch.isfinite(c[0]).all()
    with pytest.raises(RuntimeError):
        m([xm, xs, mem, xd], hx2=hx2, hx1=nz["hx1"], eps3=nz["eps3"][:49])


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["physrnn_rad", "physrad16_b", "physrad16_e3sm"])
def test_hip_radiation_graph_properties_at_shard_size(fixture):
    """2,700 columns (the per-GPU shard of the high-resolution grid): deterministic, finite, and a column's result does not
    depend on which other columns share the call (rows of a 1,200-column call = the same rows of the 2,700-column call, bit for
    bit: from 640 columns a call runs as two column halves on two streams, and the halves of both calls -- 600 and 1,350 columns --
    are in the same kernel classes: matrix-pipe GRU recurrence from 544 columns, 128 / 64-row GEMM tiles with identical k order)."""
    g, P = _load(fixture)
    m = _hip_model(P, 2700)
    B = 2700
    xm, xs, mem, xd = inputs_rad(P, B, 4242)
    nh = P["rnn2.weight_hh_l0"].shape[1]
    gen = torch.Generator().manual_seed(7)
    kw = {"hx2": torch.randn(B, nh, generator=gen)}
    if "rnn3.weight_ih" in P:
        kw.update(hx1=torch.randn(B, nh, generator=gen), eps3=torch.randn(50, B, nh, generator=gen))
    a = _run(m, xm, xs, mem, xd, **kw)
    b = _run(m, xm, xs, mem, xd, **kw)
    assert all(torch.equal(u, v) for u, v in zip(a, b)) and all(torch.isfinite(u).all() for u in a)
    n = 1200
    sub = {k: (v[:, :n] if k == "eps3" else v[:n]).contiguous() for k, v in kw.items()}
    c = _run(m, xm[:n].contiguous(), xs[:n].contiguous(), mem[:n].contiguous(), xd[:n].contiguous(), **sub)
    for u, v in zip(a, c):
        assert torch.equal(u[:n], v)
    # night columns: no shortwave reaches the surface; precipitation is never negative
    night = (xs[:, 6] * P["xdiv_sca"][6] + P["xmean_sca"][6]) < 1e-6
    assert night.any() and torch.all(a[1][night][:, [0, 4, 5, 6, 7]] == 0) and torch.all(a[1][:, 3] >= 0)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["physrnn_rad", "physrad16_b"])
def test_hip_radiation_graph_call_can_be_captured_by_the_caller(fixture):
    """A call allocates no device memory of its own and only launches on the caller's stream (the stochastic layer's side of it
    included), so a host can capture it into a hipGraph -- torch.cuda.CUDAGraph over the ctypes call -- and replay it over
    persistent buffers: replays are bit-identical to eager calls and see new buffer contents."""
    g, P = _load(fixture)
    m = _hip_model(P, 64)
    B = 48
    xm, xs, mem, xd = inputs_rad(P, B, 9)
    if getattr(m, "physrad", False):
        mem = mem.transpose(0, 1).contiguous()
    nh = P["rnn2.weight_hh_l0"].shape[1]
    gen = torch.Generator().manual_seed(3)
    kw = {"hx2": torch.randn(B, nh, generator=gen).cuda()}
    if "rnn3.weight_ih" in P:
        kw.update(hx1=torch.randn(B, nh, generator=gen).cuda(), eps3=torch.randn(50, B, nh, generator=gen).cuda())
    args = [xm.cuda(), xs.cuda(), mem.cuda(), xd.cuda()]
    ref = [t.clone() for t in m(args, **kw)]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = m(args, **kw)
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, ref))
    args[2].mul_(0.5)                                # same pointers, new memory contents: the replay must see them
    graph.replay()
    torch.cuda.synchronize()
    new = [t.clone() for t in out]
    eager = m(args, **kw)
    assert all(torch.equal(a, b) for a, b in zip(new, eager)) and not torch.equal(new[0], ref[0])
    del graph
