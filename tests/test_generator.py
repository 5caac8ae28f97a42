"""Training-data pipeline (training twin of row a1 + target construction, rnn/utils.py:2238-2371): HIP kernel vs the
numpy restatement, and the restatement vs the PINNED wrapper pre-processing for the v4 defaults."""
import numpy as np
import pytest
import torch

from conftest import load_npz_model, rel_err
from synth import synth_inputs
from oracle import generator_ref, torch_ref


def _chunk(consts, nt=3, nloc=7, seed=5, nx_sfc_in=24):
    g = np.random.Generator(np.random.PCG64(seed))
    xm, xs = synth_inputs(consts, nt * nloc, seed)
    xm[:, :, 0] = np.linspace(170.0, 310.0, 60, dtype=np.float32)[None, :] + 0.01 * xm[:, :, 0]
    if nx_sfc_in == 24:
        xs24 = np.zeros((nt * nloc, 24), np.float32)
        xs24[:, :17] = xs[:, :17]
        xs24[:, 17:22] = g.standard_normal((nt * nloc, 5)).astype(np.float32)   # the five removed past-state scalars
        xs24[:, 22:] = xs[:, 17:]
        xs24[0, 22] = 3.0e10                                                    # snow/ice sentinel
    else:
        xs24 = xs.copy()
        xs24[0, 17] = 3.0e10
    xm[1, 5, 4] = np.nan
    y = (g.standard_normal((nt * nloc, 60, 6)) * np.array([1e-5, 1e-8, 1e-9, 1e-9, 1e-5, 1e-5])).astype(np.float32)
    ys = (g.random((nt * nloc, 8)) * 1e-6).astype(np.float32)
    sh = lambda a: a.reshape((nt, nloc) + a.shape[1:])
    return {"input_lev": sh(xm), "input_sca": sh(xs24), "output_lev": sh(y), "output_sca": sh(ys)}


def _coeffs(consts, nx, ny, seed=2):
    g = np.random.Generator(np.random.PCG64(seed))
    xm, xd = consts["xmean_lev"], consts["xdiv_lev"].copy()
    xd[xd == 0] = 1.0
    if nx == 16:
        qmean = np.geomspace(2e-6, 8e-3, 60).astype(np.float32)[:, None]
        xm, xd = np.concatenate([xm, qmean], 1), np.concatenate([xd, 4 * qmean], 1)
    ys = (10 ** g.uniform(3, 7, (60, ny))).astype(np.float32)
    return ((xm, xd), (consts["xmean_sca"], consts["xdiv_sca"])), (ys, consts["yscale_sca"])


VARIANTS = [
    dict(mp_mode=1, remove_past_sfc_inputs=True),
    dict(mp_mode=0, remove_past_sfc_inputs=True, include_q_input=True, output_prune=True),
    dict(mp_mode=-1, remove_past_sfc_inputs=True, rh_input_to_q=True, rh_prune=True, qinput_prune=True),
    dict(mp_mode=-2, remove_past_sfc_inputs=True, include_q_input=True, v4_to_v5_inputs=True),
    dict(mp_mode=1, remove_past_sfc_inputs=False, cld_inp_transformation="sqrt", snowhice_fix=False, nx_sfc_in=19),
]


def _setup(kw):
    kw = dict(kw)
    consts, _, _ = load_npz_model("cur_lstm128")
    grid = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "grid_consts.npz"))
    data = _chunk(consts, nx_sfc_in=kw.pop("nx_sfc_in", 24))
    nx = 16 if kw.get("include_q_input") else 15
    ny = 5 if kw["mp_mode"] > 0 else 6
    xco, yco = _coeffs(consts, nx, ny)
    if not kw.get("remove_past_sfc_inputs", False) and data["input_sca"].shape[-1] == 24:
        raise AssertionError("test setup: 24 scalars need remove_past_sfc_inputs")
    full = dict(xcoeffs=xco, ycoeffs=yco, lbd_qc=consts["lbd_qc"], lbd_qi=consts["lbd_qi"], lbd_qn=grid["lbd_qn"],
                hyam=consts["hyam"], hybm=consts["hybm"], **kw)
    return consts, data, full


def test_restatement_input_side_equals_pinned_wrapper_preprocessing():
    """v4 defaults: generator inputs == wrapper pre-processing (pinned by the shipped TorchScript artefacts)."""
    consts, data, full = _setup(VARIANTS[0])
    xco = ((consts["xmean_lev"], consts["xdiv_lev"]), (consts["xmean_sca"], consts["xdiv_sca"]))
    full["xcoeffs"] = xco
    idx = [0, 2]
    with np.errstate(all="ignore"):
        out = generator_ref.getitem(data["input_lev"][idx].reshape(-1, 60, 15), data["input_sca"][idx].reshape(-1, 24),
                                    data["output_lev"][idx].reshape(-1, 60, 6), data["output_sca"][idx].reshape(-1, 8), **full)
    _, weights, _ = load_npz_model("cur_lstm128")
    ref = torch_ref.EmulatorRef(consts, weights, legacy=False, use_lstm=True, snowhice_fix=True)
    xs19 = np.delete(data["input_sca"][idx].reshape(-1, 24), (17, 18, 19, 20, 21), axis=1)
    xn, xsn = ref.preprocess(torch.from_numpy(data["input_lev"][idx].reshape(-1, 60, 15).copy()), torch.from_numpy(xs19))
    finite = np.isfinite(xn.numpy())
    assert rel_err(np.where(finite, out[0], 0), np.where(finite, xn.numpy(), 0)) <= 1e-6
    assert rel_err(out[1], xsn.numpy()) <= 1e-6
    assert out[2].shape == (14, 60, 5) and out[5].shape == (14, 60, 6)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", VARIANTS)
def test_hip_generator_matches_numpy_restatement(kw):
    from climsim_amd.generator import generator_xy
    consts, data, full = _setup(kw)
    gen = generator_xy(data, nloc=7, **full)
    idx = [0, 2]
    got = gen[idx]
    sh = lambda k, *s: data[k][idx].reshape(-1, *s)
    with np.errstate(all="ignore"):
        ref = generator_ref.getitem(sh("input_lev", 60, 15), sh("input_sca", data["input_sca"].shape[-1]), sh("output_lev", 60, 6),
                                    sh("output_sca", 8), **full)
    names = ["x_lev", "x_sfc", "y_lev", "y_sfc", "x_lev_denorm", "y_lev_denorm", "y_sfc_denorm"]
    for name, a, b in zip(names, got, ref):
        a = a.cpu().numpy()
        assert a.shape == b.shape, name
        assert np.array_equal(np.isnan(a), np.isnan(b)), name
        assert np.array_equal(np.isinf(a), np.isinf(b)), name
        m = np.isfinite(b)
        if a.ndim == 3:
            for v in range(a.shape[2]):   # per variable: magnitudes span many decades
                assert rel_err(np.where(m, a, 0)[:, :, v], np.where(m, b, 0)[:, :, v]) <= 2e-6, (name, v)
        else:
            assert rel_err(np.where(m, a, 0), np.where(m, b, 0)) <= 2e-6, name
    assert len(gen) == 3 * 7


@pytest.mark.gpu
def test_previous_step_inputs_and_outputs_are_appended():
    """include_prev_outputs / include_prev_inputs (rnn/utils.py:2242-2297): 5 + 6 extra level inputs from time step t-1."""
    from climsim_amd.generator import generator_xy
    consts, data, full = _setup(dict(mp_mode=1, remove_past_sfc_inputs=True))
    g = np.random.Generator(np.random.PCG64(3))
    nx = 15 + 5 + 6
    xm = np.concatenate([full["xcoeffs"][0][0], g.standard_normal((60, 11)).astype(np.float32)], 1)
    xd = np.concatenate([full["xcoeffs"][0][1], (1 + g.random((60, 11))).astype(np.float32)], 1)
    full["xcoeffs"] = ((xm, xd), full["xcoeffs"][1])
    gen = generator_xy(data, nloc=7, include_prev_inputs=True, include_prev_outputs=True, **full)
    assert gen.nx == nx and gen.ntimesteps == 2
    idx, prev = [1, 2], [0, 1]
    got = gen[idx]
    x_lev = np.concatenate([data["input_lev"][idx], data["output_lev"][prev][..., 0:5], data["input_lev"][prev][..., 0:6]], -1)
    with np.errstate(all="ignore"):
        ref = generator_ref.getitem(x_lev.reshape(-1, 60, nx), data["input_sca"][idx].reshape(-1, 24),
                                    data["output_lev"][idx].reshape(-1, 60, 6), data["output_sca"][idx].reshape(-1, 8), **full)
    for a, b in zip(got, ref):
        a = a.cpu().numpy()
        m = np.isfinite(b)
        assert a.shape == b.shape and np.array_equal(np.isfinite(a), m)
        if a.ndim == 3:
            for v in range(a.shape[2]):
                assert rel_err(np.where(m, a, 0)[:, :, v], np.where(m, b, 0)[:, :, v]) <= 2e-6, v
        else:
            assert rel_err(np.where(m, a, 0), np.where(m, b, 0)) <= 2e-6
    with pytest.raises(NotImplementedError):
        gen[[0, 1]]
