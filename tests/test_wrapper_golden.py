"""The current-generation deployment wrapper `model_wrapper` (rnn/utils.py:72-295) -- pre-processing branches, the
RH -> specific-humidity options (rows a1, a2 of SURVEY section 8) and the v5 inputs -- pinned by outputs of the reference's
OWN class, run in the build container by tests/golden/make_golden_wrapper.py:

    wrap_v4    snowhice_fix + qinput_prune + rh_prune, a 1e10 missing-value marker, a NaN and an Inf input   utils.py:182-217
    wrap_qin   include_q_input (q appended as 16th level input)                                              utils.py:262-270
    wrap_rh2q  rh_to_q (q replaces RH)                                                                       utils.py:271-272
    wrap_v5    v5_input (qn with lbd_qn, liquid fraction from T)                                             utils.py:186-198
    wrap_v5p   v5_input + qinput_prune (prune BEFORE the transform)                                          utils.py:188-190

CPU: both restatements of the oracle (C, torch) against the goldens.  GPU: the HIP path through the C ABI against them."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz_model, rel_err
from oracle import torch_ref
from oracle.pyoracle import OracleModel

TAGS = ["wrap_v4", "wrap_qin", "wrap_rh2q", "wrap_v5", "wrap_v5p"]


def _kw(flags):
    g = lambda k: bool(flags.get(k, 0))
    return dict(legacy=False, use_lstm=True, output_prune=True, scrub_inf=True, snowhice_fix=g("snowhice_fix"),
                qinput_prune=g("qinput_prune"), rh_prune=g("rh_prune"), v5_input=g("v5_input"),
                q_input_mode=1 if g("include_q_input") else (2 if g("rh_to_q") else 0))


def _steps(io):
    for B in (3, 17):
        for t in range(int(io[f"B{B}.nsteps"])):
            yield B, f"B{B}.t{t}."


def _check_outputs(o6, osf, mo, io, p, tol=1e-5):
    for v in range(6):
        assert rel_err(o6[:, :, v], io[p + "out_lev"][:, :, v]) <= tol, (p, v)
    assert rel_err(osf, io[p + "out_sfc"]) <= tol, p
    assert rel_err(mo, io[p + "mem_out"]) <= tol, p


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_restatements_vs_reference_wrapper(tag):
    consts, weights, flags = load_npz_model(tag)
    io = np.load(os.path.join(GOLDEN, f"{tag}_io.npz"))
    kw = _kw(flags)
    om = OracleModel(consts, weights, **kw)
    ref = torch_ref.EmulatorRef(consts, weights, **kw)
    for B, p in _steps(io):
        xm, xs, mem = io[p + "x_main"], io[p + "x_sfc"], io[p + "mem_in"]
        # pre-processing alone: per level input, the reference's own normalised tensors
        xn, xsn = om.preprocess(xm, xs)
        with torch.no_grad():
            tq = ref.apply_q_input(torch.from_numpy(xm), torch.from_numpy(xs))
            tn, tsn = ref.preprocess(tq, torch.from_numpy(xs))
        for v in range(xn.shape[2]):
            g = io[p + "x_main_n"][:, :, v]
            assert rel_err(xn[:, :, v], g) <= 2e-6, (p, v)
            assert rel_err(tn.numpy()[:, :, v], g) <= 2e-6, (p, v)
        assert rel_err(xsn, io[p + "x_sfc_n"]) <= 1e-6 and rel_err(tsn.numpy(), io[p + "x_sfc_n"]) <= 1e-6
        if p + "q" in io.files:       # the humidity conversion itself, element by element (8th-order Horner in fp32)
            col = 15 if kw["q_input_mode"] == 1 else 1
            q = io[p + "q"]
            back = xn[:, :, col] * consts["xdiv_lev"][:, col] + consts["xmean_lev"][:, col]
            assert np.allclose(back, q, rtol=2e-5, atol=1e-9)
        _check_outputs(*om.wrapper_forward_tuple(xm, xs, mem), io, p)
        with torch.no_grad():
            t6, tsf, tmo = ref.wrapper_forward_tuple(torch.from_numpy(xm), torch.from_numpy(xs), torch.from_numpy(mem))
        _check_outputs(t6.numpy(), tsf.numpy(), tmo.numpy(), io, p)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_wrapper_vs_reference_wrapper(tag):
    import climsim_amd
    consts, weights, flags = load_npz_model(tag)
    io = np.load(os.path.join(GOLDEN, f"{tag}_io.npz"))
    g = lambda k: bool(flags.get(k, 0))
    wrap = climsim_amd.model_wrapper(consts, weights, use_lstm=True, output_prune=True, snowhice_fix=g("snowhice_fix"),
                                     qinput_prune=g("qinput_prune"), rh_prune=g("rh_prune"), rh_to_q=g("rh_to_q"),
                                     include_q_input=g("include_q_input"), v5_input=g("v5_input"), max_batch=32)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for B, p in _steps(io):
        xm, xs = d(io[p + "x_main"]), d(io[p + "x_sfc"])
        keep = xm.clone()
        h6, hsf, hmo = wrap(xm, xs, d(io[p + "mem_in"]))
        _check_outputs(h6.cpu().numpy(), hsf.cpu().numpy(), hmo.cpu().numpy(), io, p)
        assert torch.equal(xm.isnan(), keep.isnan()) and torch.equal(torch.nan_to_num(xm), torch.nan_to_num(keep))   # inputs untouched
    # rollout: feed the HIP memory back (not the golden one) over the stored steps
    for B in (3, 17):
        mem = d(io[f"B{B}.t0.mem_in"])
        for t in range(int(io[f"B{B}.nsteps"])):
            p = f"B{B}.t{t}."
            h6, hsf, mem = wrap(d(io[p + "x_main"]), d(io[p + "x_sfc"]), mem)
            _check_outputs(h6.cpu().numpy(), hsf.cpu().numpy(), mem.cpu().numpy(), io, p)
