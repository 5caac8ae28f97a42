"""Stage-wise, teacher-forced parity of the six launches of a forward call (csa_debug_stage) on the shipped v4 wrappers at
the benchmark batch (B = 384): every HIP stage is fed the torch restatement's EXACT fp32 input for that stage and compared
with the restatement's output of that stage, per stage, at 1e-5 of the block maximum.

Why: the end-to-end output of the *stateless* v4 artefact is ill-conditioned in fp32 (its own fp32 output sits 1.3e-4
from an fp64 evaluation of the same weights), so the end-to-end test of that model can only be held to the artefact's own
conditioning.  Here a stage's error is the error of ONE kernel, and the fp64 evaluation of the same stage on the same fp32
input says how much of it is rounding of the reference itself.  What the stages show (profiles/r2_stagewise_parity.txt):
every feed-forward stage (prep, both projection GEMMs, head + post-processing + packing) and ONE cell step of either
recurrence agree with the artefact to <= 1e-6; the only place where 1e-5 is not reachable is the 60-step CHAIN of the
stateless model's recurrences, where the artefact's own fp32 result is 1.3e-5 (rnn1) / 4.7e-4 (rnn2) away from exact
arithmetic on the very same inputs -- the HIP kernel sits as close to the exact result as the artefact does, with the
hardware exp / rcp gates and with libm exp2f + IEEE division alike (CSA_FAST_GATES=0 build).

The restatement (oracle/torch_ref.py) reproduces the stateless artefact bit for bit and the memory artefact to 1e-6
(tests/test_oracle_golden.py), so its intermediates ARE the artefact's.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz_model
from synth import synth_inputs

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)
TOL = 1e-5
PERM = [0, 2, 1, 3]          # PyTorch gate rows (i, f, g, o) -> the library's per-unit order [i, g~, f, o]


def _rel(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / np.abs(ref).max())


def reference_stages(tag, B, seed, dtype):
    """Every intermediate of the legacy wrapper, evaluated in `dtype` FROM THE fp32 INTERMEDIATE OF THE PREVIOUS STAGE
    (teacher forcing): returns {stage: (inputs..., output)} as fp32 / fp64 numpy."""
    from oracle import torch_ref
    consts, weights, _ = load_npz_model(tag)
    r32 = torch_ref.EmulatorRef(consts, weights, legacy=True)
    r = r32 if dtype == torch.float32 else torch_ref.EmulatorRef(consts, weights, legacy=True, dtype=dtype)
    xm, xs = synth_inputs(consts, B, seed)
    g = np.random.Generator(np.random.PCG64(seed))
    hx2, cx2 = g.standard_normal((2, B, r.nh2)).astype(np.float32)
    mem = (0.3 * g.standard_normal((B, 60, 16))).astype(np.float32) if r.nh_mem else None
    t = lambda a: None if a is None else torch.from_numpy(np.asarray(a))
    c = lambda a: a.to(dtype)
    L = r.nlev
    out = {"inputs": (xm, xs, mem, hx2, cx2)}
    with torch.no_grad():
        # ---- the fp32 chain (what the artefact computes), each piece kept -------------------------------------------
        def prep(m, x_main, x_sfc, mem_in):
            xn, xsn = m.preprocess(x_main, x_sfc)
            x = xn.transpose(0, 1)
            sp = xsn[:, 0:1] * m.xdiv_sca[0] + m.xmean_sca[0]
            pres = torch.sqrt(m.hyam.view(-1, 1, 1) * 100000.0 + sp.unsqueeze(0) * m.hybm.view(-1, 1, 1)) / 314.0
            x = torch.flip(torch.tanh(m.mlp_initial(torch.cat((x, pres), dim=2))), [0])
            if m.nh_mem:
                x = torch.cat((x, mem_in.transpose(0, 1)), dim=2)
            return x, torch.tanh(m.mlp_surface1(xsn)), torch.tanh(m.mlp_surface2(xsn))

        def proj(rnn, X):
            return X @ rnn.weight_ih_l0.t() + rnn.bias_ih_l0 + rnn.bias_hh_l0

        def rec(rnn, X, h0, c0):
            """ATen's LSTM, one level at a time so that (h_{t-1}, c_{t-1}) of every step is available: (H, Hprev, Cprev)."""
            h, c_, hs, hp, cp = h0.unsqueeze(0), c0.unsqueeze(0), [], [], []
            for k in range(X.shape[0]):
                hp.append(h[0]); cp.append(c_[0])
                o, (h, c_) = rnn(X[k:k + 1], (h, c_))
                hs.append(o[0])
            return torch.stack(hs), torch.stack(hp), torch.stack(cp)

        def step(rnn, P, Hp, Cp):
            """ONE LSTM cell step per (level, column) from the given pre-activations and previous state (i, f, g, o rows)."""
            g = P + Hp @ rnn.weight_hh_l0.t()
            i, f, gg, o = g.chunk(4, dim=-1)
            cn = torch.sigmoid(f) * Cp + torch.sigmoid(i) * torch.tanh(gg)
            return torch.sigmoid(o) * torch.tanh(cn)

        def head(m, r2, x_main, B):
            last_h = r2[-1]
            if m.nh_mem:
                z = m.mlp_latent(r2)
                mem_out = torch.flip(z, [0]).transpose(0, 1)
            else:
                z, mem_out = r2, None
            o = m.mlp_output(z).transpose(0, 1)
            o6, os_ = m.postprocess(o, m.mlp_surface_output(last_h), x_main)
            parts = [o6.transpose(1, 2).reshape(B, -1), os_] + ([mem_out.reshape(B, -1)] if mem_out is not None else [])
            return torch.cat(parts, dim=1)

        X1, h0, c0 = prep(r32, t(xm), t(xs), t(mem))
        P1 = proj(r32.rnn1, X1)
        H1s, Hp1, Cp1 = rec(r32.rnn1, X1, h0, c0)                        # sequence order
        assert torch.equal(H1s, r32.rnn1(X1, (h0.unsqueeze(0), c0.unsqueeze(0)))[0])   # level-by-level == one call, bit for bit
        H1 = torch.flip(H1s, [0])                                        # level order
        P2 = proj(r32.rnn2, H1)
        H2, Hp2, Cp2 = rec(r32.rnn2, H1, t(hx2), t(cx2))
        Y = head(r32, H2, t(xm), B)
        chain = dict(X1=X1, h0=h0, c0=c0, P1=P1, H1=H1, P2=P2, H2=H2, Y=Y, Hp1=Hp1, Cp1=Cp1, Hp2=Hp2, Cp2=Cp2,
                     S1=step(r32.rnn1, P1, Hp1, Cp1), S2=step(r32.rnn2, P2, Hp2, Cp2))
        if dtype == torch.float32:
            res = chain
        else:   # same stages in fp64, each FROM the fp32 input of that stage
            X1d, h0d, c0d = prep(r, c(t(xm)), c(t(xs)), None if mem is None else c(t(mem)))
            res = dict(X1=X1d, h0=h0d, c0=c0d, P1=proj(r.rnn1, c(X1)),
                       H1=torch.flip(rec(r.rnn1, c(X1), c(h0), c(c0))[0], [0]), P2=proj(r.rnn2, c(H1)),
                       H2=rec(r.rnn2, c(H1), c(t(hx2)), c(t(cx2)))[0], Y=head(r, c(H2), c(t(xm)), B),
                       S1=step(r.rnn1, c(P1), c(Hp1), c(Cp1)), S2=step(r.rnn2, c(P2), c(Hp2), c(Cp2)))
    out.update({k: v.numpy() for k, v in res.items()})
    out["chain32"] = {k: v.numpy() for k, v in chain.items()}
    return out


def hip_stages(tag, B, ref32):
    """Each HIP launch on the fp32 reference input of its stage."""
    import climsim_amd
    consts, weights, _ = load_npz_model(tag)
    model = climsim_amd.NewModel_constraint(consts, weights, max_batch=B)
    em, cfg = model.emulator, model.emulator.cfg
    L, nh, nm = cfg.nlev, cfg.nh1, cfg.nh_mem
    d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    xm, xs, mem, hx2, cx2 = ref32["inputs"]
    ch = ref32["chain32"]
    unit_major = lambda P: np.ascontiguousarray(P.reshape(L, B, 4, nh)[:, :, PERM, :].transpose(0, 1, 3, 2)).reshape(L * B, 4 * nh)
    out = {}
    X1, hc0 = em.debug_stage(0, B, [d(xm), d(xs), d(mem), d(hx2), d(cx2)], [(L, B, nh + nm), (4, B, nh)])
    out["X1"], out["h0"], out["c0"] = X1.cpu().numpy(), hc0[0].cpu().numpy(), hc0[1].cpu().numpy()
    (P1,) = em.debug_stage(1, B, [d(ch["X1"].reshape(L * B, nh + nm))], [(L * B, 4 * nh)])
    out["P1"] = P1.cpu().numpy()
    (H1,) = em.debug_stage(2, B, [d(unit_major(ch["P1"])), d(ch["h0"]), d(ch["c0"])], [(L, B, nh)])
    out["H1"] = H1.cpu().numpy()
    (P2,) = em.debug_stage(3, B, [d(ch["H1"].reshape(L * B, nh))], [(L * B, 4 * nh)])
    out["P2"] = P2.cpu().numpy()
    (H2,) = em.debug_stage(4, B, [d(unit_major(ch["P2"])), d(hx2), d(cx2)], [(L, B, nh)])
    out["H2"] = H2.cpu().numpy()
    (Y,) = em.debug_stage(5, B, [d(ch["H2"]), d(xm), d(xs)], [(B, em.packed_width)])
    out["Y"] = Y.cpu().numpy()
    # ONE cell step per (level, column): the recurrent kernel's arithmetic without the 60-step chain
    flat = lambda a: a.reshape(L * B, nh)
    (S1,) = em.debug_stage(6, B, [d(unit_major(ch["P1"])), d(flat(ch["Hp1"])), d(flat(ch["Cp1"]))], [(L, B, nh)])
    (S2,) = em.debug_stage(7, B, [d(unit_major(ch["P2"])), d(flat(ch["Hp2"])), d(flat(ch["Cp2"]))], [(L, B, nh)])
    out["S1"], out["S2"] = S1.cpu().numpy(), S2.cpu().numpy()
    torch.cuda.synchronize()
    out["_unit_major"] = unit_major
    em.close()
    return out


def stage_errors(tag, B=384, seed=4242):
    r32 = reference_stages(tag, B, seed, torch.float32)
    r64 = reference_stages(tag, B, seed, torch.float64)
    hip = hip_stages(tag, B, r32)
    um = hip.pop("_unit_major")
    rows = []
    for k in ("X1", "h0", "c0", "P1", "S1", "H1", "P2", "S2", "H2"):
        a32, a64 = r32[k], r64[k]
        if k in ("P1", "P2"):
            a32, a64 = um(a32), um(a64)
        rows.append((k, _rel(hip[k], a32), _rel(hip[k], a64), _rel(a32, a64)))
    nmem = hip["Y"].shape[1] - 368
    for name, (a, b) in {"Y.lev": (0, 360), "Y.sfc": (360, 368), **({"Y.mem": (368, 368 + nmem)} if nmem else {})}.items():
        rows.append((name, _rel(hip["Y"][:, a:b], r32["Y"][:, a:b]), _rel(hip["Y"][:, a:b], r64["Y"][:, a:b]),
                     _rel(r32["Y"][:, a:b], r64["Y"][:, a:b])))
    return rows


NAMES = {"X1": "prep -> rnn1 input rows", "h0": "prep -> h0 (mlp_surface1)", "c0": "prep -> c0 (mlp_surface2)",
         "P1": "projection GEMM rnn1", "S1": "rnn1, ONE cell step", "H1": "rnn1, 60-step chain",
         "P2": "projection GEMM rnn2", "S2": "rnn2, ONE cell step", "H2": "rnn2, 60-step chain",
         "Y.lev": "head+post+pack: lev", "Y.sfc": "head+post+pack: sfc", "Y.mem": "head+pack: memory"}


def format_rows(tag, rows):
    out = [f"{tag}, B = 384: stage                | HIP vs torch fp32 | HIP vs fp64 | torch fp32 vs fp64   (max|a-b| / max|ref|)"]
    out += [f"  {NAMES[k]:32s} {e32:12.2e} {e64:14.2e} {r:14.2e}" for k, e32, e64, r in rows]
    return "\n".join(out)


@pytest.mark.parametrize("tag", ["v4_stateless", "v4_memory"])
def test_every_stage_within_1e5_of_the_reference_stage(tag):
    rows = stage_errors(tag)
    print("\n" + format_rows(tag, rows))
    for k, e32, e64, r in rows:
        if k in ("H1", "H2"):
            # a 60-step chain amplifies fp32 rounding; where that exceeds 1e-5 in the REFERENCE itself (column r: stateless
            # rnn1 1.3e-5, rnn2 4.7e-4; memory rnn1 2.2e-5) the HIP kernel must be as close to exact arithmetic as the
            # reference is (never worse than 2x + 1e-5); everywhere else the flat 1e-5 holds
            assert e32 <= TOL or (r > TOL and e64 <= 2.0 * r + TOL), (k, e32, e64, r)
        else:
            assert e32 <= TOL, (k, e32, e64, r)
        if k in ("X1", "h0", "c0", "P1", "P2", "Y.lev", "Y.sfc", "Y.mem"):
            assert e32 <= 2e-6, (k, e32)        # feed-forward kernels: measured <= 6e-7
        if k in ("S1", "S2"):
            assert e64 <= r + 1e-6, (k, e64, r)  # one cell step: at least as close to exact arithmetic as the fp32 reference step
