"""CPU: pin the oracle (C restatement + torch restatement) against golden vectors produced by the
reference's own artefacts / classes (tests/golden/make_golden*.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, block_errors, conditioning_tol, load_npz_model, rel_err
from synth import checksum, synth_inputs

from oracle import torch_ref
from oracle.pyoracle import OracleModel

torch.set_num_threads(4)


def _stateless_inputs(io, consts, B):
    if f"B{B}.x_main" in io.files:
        return io[f"B{B}.x_main"], io[f"B{B}.x_sfc"]
    xm, xs = synth_inputs(consts, B, int(io[f"B{B}.seed"]))
    assert checksum(xm, xs) == io[f"B{B}.x_checksum"]
    return xm, xs


@pytest.mark.parametrize("B", [1, 8, 67, 384])
def test_torch_restatement_bitexact_vs_stateless_artefact(B):
    """The torch restatement reproduces rnn/v4_rnn_wrapper_constrained.pt bit for bit (same ATen
    kernels, same op order) -> the restated algorithm IS the reference's."""
    consts, weights, _ = load_npz_model("v4_stateless")
    io = np.load(os.path.join(GOLDEN, "v4_stateless_io.npz"))
    ref = torch_ref.EmulatorRef(consts, weights, legacy=True)
    xm, xs = _stateless_inputs(io, consts, B)
    with torch.no_grad():
        y = ref.wrapper_forward(torch.from_numpy(xm), torch.from_numpy(xs), None,
                                torch.from_numpy(io[f"B{B}.hx2"]), torch.from_numpy(io[f"B{B}.cx2"])).numpy()
    g = io[f"B{B}.yout"]
    err = block_errors(y, g)
    # identical ATen kernels: allow 1e-6 in case the CPU ISA dispatch differs between hosts
    assert max(err.values()) <= 1e-6, err


@pytest.mark.parametrize("B", [1, 8, 67, 384])
def test_c_oracle_vs_stateless_artefact(B):
    """The stateless v4 model is ill-conditioned in fp32: its own fp32 output sits ~1.3e-4 (of the
    block max, B=384) from the fp64 evaluation of the same weights.  The C oracle (a different fp32
    summation order) is held to conditioning_tol() = 2x that discrepancy, not to 1e-5."""
    consts, weights, _ = load_npz_model("v4_stateless")
    io = np.load(os.path.join(GOLDEN, "v4_stateless_io.npz"))
    om = OracleModel(consts, weights, legacy=True)
    xm, xs = _stateless_inputs(io, consts, B)
    y = om.wrapper_forward(xm, xs, None, io[f"B{B}.hx2"], io[f"B{B}.cx2"])
    tol = conditioning_tol("v4_stateless")
    assert 1e-5 < tol < 1e-3
    err = block_errors(y, io[f"B{B}.yout"])
    assert max(err.values()) <= tol, (err, tol)


@pytest.mark.parametrize("B", [1, 8, 384])
def test_oracles_vs_memory_artefact_rollout(B):
    """rnn/v4_rnn-memory_wrapper_constrained_huber.pt, caller-owned memory fed back: both
    restatements within the 1e-5 block tolerance at every step."""
    consts, weights, _ = load_npz_model("v4_memory")
    io = np.load(os.path.join(GOLDEN, "v4_memory_io.npz"))
    om = OracleModel(consts, weights, legacy=True)
    ref = torch_ref.EmulatorRef(consts, weights, legacy=True)
    mem = np.zeros((B, 60, 16), np.float32)
    for t in range(int(io[f"B{B}.nsteps"])):
        p = f"B{B}.t{t}."
        if p + "x_main" in io.files:
            xm, xs = io[p + "x_main"], io[p + "x_sfc"]
            assert np.array_equal(mem, io[p + "mem_in"])
        else:
            xm, xs = synth_inputs(consts, B, int(io[p + "seed"]))
            assert checksum(xm, xs) == io[p + "x_checksum"]
        g = io[p + "yout"]
        y = om.wrapper_forward(xm, xs, mem, io[p + "hx2"], io[p + "cx2"])
        err = block_errors(y, g)
        assert max(err.values()) <= conditioning_tol("v4_memory") == 1e-5, (t, err)
        with torch.no_grad():
            yt = ref.wrapper_forward(torch.from_numpy(xm), torch.from_numpy(xs), torch.from_numpy(mem),
                                     torch.from_numpy(io[p + "hx2"]), torch.from_numpy(io[p + "cx2"])).numpy()
        err = block_errors(yt, g)
        assert max(err.values()) <= 1e-6, (t, err)
        mem = g[:, 368:].reshape(B, 60, 16).copy()


@pytest.mark.parametrize("tag", ["cur_lstm128", "cur_lstm144", "cur_gru128"])
def test_oracles_vs_current_class(tag):
    """Golden I/O of rnn/models/models.py::RNN_autoreg (+postprocessing) run in the build container."""
    consts, weights, flags = load_npz_model(tag)
    io = np.load(os.path.join(GOLDEN, f"{tag}_io.npz"))
    kw = dict(legacy=False, use_lstm=bool(flags["use_lstm"]), output_prune=bool(flags["output_prune"]),
              scrub_inf=True)
    om = OracleModel(consts, weights, **kw)
    ref = torch_ref.EmulatorRef(consts, weights, **kw)
    for B in (2, 16):
        for t in range(int(io[f"B{B}.nsteps"])):
            p = f"B{B}.t{t}."
            xn, xs = om.preprocess(io[p + "x_main"], io[p + "x_sfc"])
            assert rel_err(xn, io[p + "x_main_n"]) <= 1e-6
            assert rel_err(xs, io[p + "x_sfc_n"]) <= 1e-6
            out, out_sfc, mem_out = om.model_forward(io[p + "x_main_n"], io[p + "x_sfc_n"], io[p + "mem_in"])
            assert rel_err(out, io[p + "out"]) <= 1e-5
            assert rel_err(out_sfc, io[p + "out_sfc"]) <= 1e-5
            assert rel_err(mem_out, io[p + "mem_out"]) <= 1e-5
            o6, osd, mo = om.wrapper_forward_tuple(io[p + "x_main"], io[p + "x_sfc"], io[p + "mem_in"])
            for v in range(6):   # per-variable block: magnitudes span 12 decades
                assert rel_err(o6[:, :, v], io[p + "post_lev"][:, :, v]) <= 1e-5, v
            assert rel_err(osd / consts["yscale_sca"] * 0 + osd, io[p + "post_sfc"]) <= 1e-5
            with torch.no_grad():
                to, tos, tm = ref.model_forward(torch.from_numpy(io[p + "x_main_n"]),
                                                torch.from_numpy(io[p + "x_sfc_n"]),
                                                torch.from_numpy(io[p + "mem_in"]))
            assert rel_err(to.numpy(), io[p + "out"]) <= 1e-6
            assert rel_err(tm.numpy(), io[p + "mem_out"]) <= 1e-6
            assert rel_err(tos.numpy(), io[p + "out_sfc"]) <= 1e-6


@pytest.mark.parametrize("tag", ["cur_mpm1", "cur_mpm2", "cur_stoch"])
def test_torch_restatement_vs_current_class_variants(tag):
    """mp_mode -1 / -2 post-processing and the 3-RNN stochastic model (tests/golden/make_golden_variants.py)."""
    consts, weights, flags = load_npz_model(tag)
    io = np.load(os.path.join(GOLDEN, f"{tag}_io.npz"))
    ref = torch_ref.EmulatorRef(consts, weights, legacy=False, use_lstm=True, output_prune=bool(flags["output_prune"]),
                                mp_mode=int(flags["mp_mode"]), scrub_inf=True)
    T = torch.from_numpy
    for B in (3, 10):
        for t in range(int(io[f"B{B}.nsteps"])):
            p = f"B{B}.t{t}."
            noise = tuple(T(io[p + k]) for k in ("hx0", "cx0", "eps")) if tag == "cur_stoch" else None
            with torch.no_grad():
                xn, xs = ref.preprocess(T(io[p + "x_main"]), T(io[p + "x_sfc"]))
                assert rel_err(xn.numpy(), io[p + "x_main_n"]) <= 1e-6
                to, tos, tm = ref.model_forward(T(io[p + "x_main_n"]), T(io[p + "x_sfc_n"]), T(io[p + "mem_in"]), noise=noise)
                assert rel_err(to.numpy(), io[p + "out"]) <= 2e-6
                assert rel_err(tm.numpy(), io[p + "mem_out"]) <= 2e-6
                assert rel_err(tos.numpy(), io[p + "out_sfc"]) <= 2e-6
                o6, osd = ref.postprocess(T(io[p + "out"]), T(io[p + "out_sfc"]), T(io[p + "x_main"]))
            for v in range(6):
                assert rel_err(o6.numpy()[:, :, v], io[p + "post_lev"][:, :, v]) <= 1e-6, v
            assert rel_err(osd.numpy(), io[p + "post_sfc"]) <= 1e-6


def test_preprocess_edge_cases():
    """NaN/Inf scrub, snow/ice sentinel, zero divisors (xdiv has 66 exact zeros), RH clamp, q prune."""
    consts, weights, _ = load_npz_model("v4_memory")
    xm, xs = synth_inputs(consts, 4, 5)
    xm[0, 3, 0] = np.nan
    xm[1, 30, 13] += 1.0          # xdiv == 0 there -> +Inf
    xm[2, 5, 1] = 7.0             # RH far above the clamp
    xs[3, 15] = 1e30              # snow/ice sentinel
    for flags in (dict(), dict(scrub_inf=True, snowhice_fix=True, rh_prune=True, qinput_prune=True)):
        om = OracleModel(consts, weights, legacy=True, **flags)
        ref = torch_ref.EmulatorRef(consts, weights, legacy=True, **flags)
        a, b = om.preprocess(xm, xs)
        with torch.no_grad():
            ta, tb = ref.preprocess(torch.from_numpy(xm), torch.from_numpy(xs))
        assert np.array_equal(np.isinf(a), np.isinf(ta.numpy()))
        m = np.isfinite(a)
        assert np.allclose(a[m], ta.numpy()[m], rtol=2e-6, atol=1e-7)
        assert np.allclose(b, tb.numpy(), rtol=2e-6, atol=1e-30)
        assert not np.isnan(a).any()
        if flags:
            assert not np.isinf(a).any()
            assert a[2, 5, 1] == np.float32(1.2) and (a[:, :15, 2] == 0).all()
            assert b[3, 15] == (np.float32(-1.0) - consts["xmean_sca"][15]) / consts["xdiv_sca"][15]
        else:
            assert np.isinf(a[1, 30, 13])
