import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz_model(tag):
    d = np.load(os.path.join(GOLDEN, f"{tag}_model.npz"))
    consts = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
    weights = {k[2:]: d[k] for k in d.files if k.startswith("w.")}
    flags = {k[6:]: int(d[k]) for k in d.files if k.startswith("flags.")}
    return consts, weights, flags


def block_errors(y, ref, nlev=60, ny_sfc=8):
    """SURVEY section 7 / BASELINE.md section 4 tolerance definition: max|a-b| / max|ref| per output
    block (lev 0:360, sfc 360:368, mem 368:).  NaN patterns must coincide."""
    out = {}
    n_lev = 6 * nlev
    for a, b, name in ((0, n_lev, "lev"), (n_lev, n_lev + ny_sfc, "sfc"), (n_lev + ny_sfc, None, "mem")):
        r = ref[:, a:b]
        if r.size == 0:
            continue
        v = y[:, a:b]
        assert np.array_equal(np.isnan(r), np.isnan(v)), f"NaN pattern differs in block {name}"
        m = np.isfinite(r)
        out[name] = float(np.abs(v[m] - r[m]).max() / np.abs(r[m]).max())
    return out


def rel_err(a, ref):
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-300))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_COND = {}


def conditioning_tol(tag="v4_stateless"):
    """fp32 conditioning of a shipped model: the reference's OWN fp32 output (golden, B=384) is
    compared with an fp64 evaluation of the same weights/inputs; an independent fp32
    implementation cannot be expected to sit closer to the reference than the reference sits
    to exact arithmetic.  Returns max(1e-5, 2 x that discrepancy) -- 1e-5 for the
    well-conditioned memory wrapper, ~2.6e-4 for the ill-conditioned stateless one."""
    if tag in _COND:
        return _COND[tag]
    import torch
    from synth import synth_inputs
    from oracle import torch_ref
    consts, weights, _ = load_npz_model(tag)
    io = np.load(os.path.join(GOLDEN, f"{tag}_io.npz"))
    r64 = torch_ref.EmulatorRef(consts, weights, legacy=True, dtype=torch.float64)
    d = lambda a: torch.from_numpy(np.asarray(a)).double()
    if tag == "v4_stateless":
        xm, xs = synth_inputs(consts, 384, int(io["B384.seed"]))
        with torch.no_grad():
            y64 = r64.wrapper_forward(d(xm), d(xs), None, d(io["B384.hx2"]), d(io["B384.cx2"])).numpy()
        g = io["B384.yout"]
    else:
        xm, xs = synth_inputs(consts, 384, int(io["B384.t0.seed"]))
        with torch.no_grad():
            y64 = r64.wrapper_forward(d(xm), d(xs), d(np.zeros((384, 60, 16))), d(io["B384.t0.hx2"]),
                                      d(io["B384.t0.cx2"])).numpy()
        g = io["B384.t0.yout"]
    e = max(block_errors(g, y64).values())
    _COND[tag] = max(1e-5, 2.0 * e)
    return _COND[tag]
