"""Training-data pipeline (training twin of row a1 + target construction, rnn/utils.py:1870-2384): the HIP kernel and the numpy
restatement against outputs of the REFERENCE CLASS ITSELF -- generator_xy's own constructor and `__getitem__`, run in the build
container on in-memory datasets by tests/golden/make_golden_generator.py (generator_golden.npz: mp_mode 0 / 1 / -1 / -2 targets, the
float64 RH -> q conversion, v4_to_v5_inputs, the numba normalisers, re-normalisation, previous-step inputs / outputs)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz_model, rel_err
from synth import GENERATOR_VARIANTS, generator_setup
from oracle import generator_ref, torch_ref

NAMES = ["x_lev", "x_sfc", "y_lev", "y_sfc", "x_lev_denorm", "y_lev_denorm", "y_sfc_denorm"]


def _load():
    consts, _, _ = load_npz_model("cur_lstm128")
    grid = np.load(os.path.join(GOLDEN, "grid_consts.npz"))
    return consts, grid["lbd_qn"], np.load(os.path.join(GOLDEN, "generator_golden.npz"))


def _compare(got, gold, tag, tol=2e-6):
    for name, a in zip(NAMES, got):
        b = gold[f"{tag}.{name}"]
        a = a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        assert a.shape == b.shape, (tag, name, a.shape, b.shape)
        assert np.array_equal(np.isnan(a), np.isnan(b)), (tag, name)
        assert np.array_equal(np.isinf(a), np.isinf(b)) and np.array_equal(np.sign(a[np.isinf(a)]), np.sign(b[np.isinf(b)])), (tag, name)
        m = np.isfinite(b)
        if a.ndim == 3:
            for v in range(a.shape[2]):   # per variable: magnitudes span many decades
                assert rel_err(np.where(m, a, 0)[:, :, v], np.where(m, b, 0)[:, :, v]) <= tol, (tag, name, v)
        else:
            assert rel_err(np.where(m, a, 0), np.where(m, b, 0)) <= tol, (tag, name)


def _restatement(data, full, idx):
    sh = lambda k: data[k][idx].reshape((-1,) + data[k].shape[2:])
    x_lev = sh("input_lev")
    if full.get("include_prev_inputs") or full.get("include_prev_outputs"):
        prev = [idx[0] - 1] + idx[:-1]
        x_lev = np.concatenate([data["input_lev"][idx], data["output_lev"][prev][..., 0:5], data["input_lev"][prev][..., 0:6]], -1)
        x_lev = x_lev.reshape((-1,) + x_lev.shape[2:])
    kw = {k: v for k, v in full.items() if k not in ("include_prev_inputs", "include_prev_outputs")}
    with np.errstate(all="ignore"):
        return generator_ref.getitem(x_lev, sh("input_sca"), sh("output_lev"), sh("output_sca"), **kw)


@pytest.mark.parametrize("tag", list(GENERATOR_VARIANTS))
def test_restatement_vs_reference_class(tag):
    """oracle/generator_ref.py pinned by generator_xy.__getitem__ itself, every variant."""
    consts, lbd_qn, gold = _load()
    data, full = generator_setup(consts, lbd_qn, tag)
    got = _restatement(data, full, [int(i) for i in gold[f"{tag}.idx"]])
    _compare(got, gold, tag, tol=1e-6)


def test_float64_humidity_conversion_vs_reference_functions():
    """eliq / eice / relative_to_specific_humidity_climsim (rnn/utils.py:647-690) over all three branches of eice, float64."""
    _, _, gold = _load()
    T, rh, p = gold["rh2q.T"], gold["rh2q.rh"], gold["rh2q.p"]
    assert np.abs(generator_ref.eliq(T) / gold["rh2q.eliq"] - 1).max() <= 1e-14
    assert np.abs(generator_ref.eice(T) / gold["rh2q.eice"] - 1).max() <= 1e-14
    q = generator_ref.rh_to_q(rh, T, p)
    assert np.abs(q - gold["rh2q.q"]).max() <= 1e-14 * np.abs(gold["rh2q.q"]).max()


def test_restatement_input_side_equals_pinned_wrapper_preprocessing():
    """v4 defaults: generator inputs == wrapper pre-processing (pinned by the shipped TorchScript artefacts and model_wrapper)."""
    consts, lbd_qn, _ = _load()
    data, full = generator_setup(consts, lbd_qn, "mp1")
    full["xcoeffs"] = ((consts["xmean_lev"], consts["xdiv_lev"]), (consts["xmean_sca"], consts["xdiv_sca"]))
    idx = [0, 2]
    out = _restatement(data, full, idx)
    _, weights, _ = load_npz_model("cur_lstm128")
    ref = torch_ref.EmulatorRef(consts, weights, legacy=False, use_lstm=True, snowhice_fix=True)
    xs19 = np.delete(data["input_sca"][idx].reshape(-1, 24), (17, 18, 19, 20, 21), axis=1)
    xn, xsn = ref.preprocess(torch.from_numpy(data["input_lev"][idx].reshape(-1, 60, 15).copy()), torch.from_numpy(xs19))
    finite = np.isfinite(xn.numpy())
    assert rel_err(np.where(finite, out[0], 0), np.where(finite, xn.numpy(), 0)) <= 1e-6
    assert rel_err(out[1], xsn.numpy()) <= 1e-6
    assert out[2].shape == (14, 60, 5) and out[5].shape == (14, 60, 6)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(GENERATOR_VARIANTS))
def test_hip_generator_vs_reference_class(tag):
    """gen.hip behind climsim_amd.generator.generator_xy (the reference's constructor keywords and return order) against the
    reference class's own outputs on the same in-memory datasets."""
    from climsim_amd.generator import generator_xy
    consts, lbd_qn, gold = _load()
    data, full = generator_setup(consts, lbd_qn, tag)
    gen = generator_xy(data, nloc=7, **full)
    assert [gen.nx, gen.nx_sfc, gen.ny, gen.ny_sfc] == [int(v) for v in gold[f"{tag}.dims"]]
    prev = full.get("include_prev_inputs") or full.get("include_prev_outputs")
    assert len(gen) == (2 if prev else 3) * 7
    _compare(gen[[int(i) for i in gold[f"{tag}.idx"]]], gold, tag)
    if prev:
        with pytest.raises(NotImplementedError):
            gen[[0, 1]]
