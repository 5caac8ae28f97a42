"""RH -> specific-humidity input options of the current wrapper (row a2, rnn/utils.py:134-180, 262-272):
include_q_input (q appended as 16th level input) and rh_to_q (q replaces RH).

Pinning: rnn/utils.py is not importable here (numba/h5py absent) and no shipped artefact contains the v4
wrapper with these options, so this arithmetic is pinned by RESTATEMENT only: two independent restatements
(C oracle, torch) must agree, and the HIP path must agree with them -- "parity unpinned" by reference outputs."""
import numpy as np
import pytest
import torch

from conftest import load_npz_model, rel_err
from synth import synth_inputs
from oracle import torch_ref
from oracle.pyoracle import OracleModel


def _model_with_q(mode):
    consts, weights, flags = load_npz_model("cur_lstm128")
    consts = dict(consts)
    weights = dict(weights)
    if mode == 1:      # 16 level inputs: append normalisation constants and a weight column for q
        g = np.random.Generator(np.random.PCG64(5))
        qmean = np.geomspace(2e-6, 8e-3, 60).astype(np.float32)[:, None]
        consts["xmean_lev"] = np.concatenate([consts["xmean_lev"], qmean], axis=1)
        consts["xdiv_lev"] = np.concatenate([consts["xdiv_lev"], 4 * qmean], axis=1)
        w = weights["mlp_initial.weight"]                      # (128, 16) = 15 inputs + pressure
        extra = (0.3 * g.standard_normal((w.shape[0], 1))).astype(np.float32)
        weights["mlp_initial.weight"] = np.concatenate([w[:, :15], extra, w[:, 15:]], axis=1)
    return consts, weights, flags


@pytest.mark.parametrize("mode", [1, 2])
def test_c_and_torch_restatements_agree(mode):
    consts, weights, flags = _model_with_q(mode)
    kw = dict(legacy=False, use_lstm=True, output_prune=True, scrub_inf=True, snowhice_fix=True, q_input_mode=mode)
    om = OracleModel(consts, weights, **kw)
    ref = torch_ref.EmulatorRef(consts, weights, **kw)
    B = 9
    xm, xs = synth_inputs({k: v[:, :15] if k in ("xmean_lev", "xdiv_lev") else v for k, v in consts.items()}, B, 3)
    xm[:, :, 0] = np.linspace(170.0, 310.0, 60, dtype=np.float32)[None, :] + xm[:, :, 0] * 0.01   # all three e_ice branches
    mem = (0.2 * np.random.default_rng(1).standard_normal((60, B, 16))).astype(np.float32)
    o6, osf, mo = om.wrapper_forward_tuple(xm, xs, mem)
    with torch.no_grad():
        t6, tsf, tmo = ref.wrapper_forward_tuple(torch.from_numpy(xm), torch.from_numpy(xs), torch.from_numpy(mem))
    for v in range(6):
        assert rel_err(o6[:, :, v], t6.numpy()[:, :, v]) <= 1e-5, v
    assert rel_err(osf, tsf.numpy()) <= 1e-5
    assert rel_err(mo, tmo.numpy()) <= 1e-5
    # the q column itself (normalised) to fp32 rounding
    xn, _ = om.preprocess(xm, xs)
    with torch.no_grad():
        xq = ref.apply_q_input(torch.from_numpy(xm), torch.from_numpy(xs))
        tn, _ = ref.preprocess(xq, torch.from_numpy(xs))
    col = 15 if mode == 1 else 1
    assert rel_err(xn[:, :, col], tn.numpy()[:, :, col]) <= 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_hip_q_input_matches_oracle(mode):
    import climsim_amd
    consts, weights, flags = _model_with_q(mode)
    om = OracleModel(consts, weights, legacy=False, use_lstm=True, output_prune=True, scrub_inf=True, snowhice_fix=True,
                     q_input_mode=mode)
    wrap = climsim_amd.model_wrapper(consts, weights, use_lstm=True, output_prune=True, snowhice_fix=True,
                                     include_q_input=(mode == 1), rh_to_q=(mode == 2), max_batch=16)
    B = 9
    xm, xs = synth_inputs({k: v[:, :15] if k in ("xmean_lev", "xdiv_lev") else v for k, v in consts.items()}, B, 3)
    xm[:, :, 0] = np.linspace(170.0, 310.0, 60, dtype=np.float32)[None, :] + xm[:, :, 0] * 0.01
    mem = (0.2 * np.random.default_rng(1).standard_normal((60, B, 16))).astype(np.float32)
    o6, osf, mo = om.wrapper_forward_tuple(xm, xs, mem)
    d = lambda a: torch.from_numpy(a).cuda()
    h6, hsf, hmo = wrap(d(xm), d(xs), d(mem))
    for v in range(6):
        assert rel_err(h6.cpu().numpy()[:, :, v], o6[:, :, v]) <= 1e-5, v
    assert rel_err(hsf.cpu().numpy(), osf) <= 1e-5
    assert rel_err(hmo.cpu().numpy(), mo) <= 1e-5
