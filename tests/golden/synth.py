"""Seeded synthetic column inputs shared by make_golden.py, the tests and bench.py.

Recipe from SURVEY.md section 8(d): x_main = xmean_lev + xdiv_lev*U(-.5,.5) with
RH (var 1) ~ U(0,1) and qliq/qice (vars 2,3) ~ 1e-4*U(0,1); x_sfc = xmean_sca +
xdiv_sca*U(-.5,.5).  numpy's PCG64 stream is version-stable, so large-B inputs
are regenerated from the seed instead of being committed.
"""
import numpy as np


def synth_inputs(consts, B, seed, nlev=60, nx=15, nx_sfc=19):
    g = np.random.Generator(np.random.PCG64(int(seed)))
    xm, xd = consts["xmean_lev"], consts["xdiv_lev"]
    x_main = (xm[None] + xd[None] * g.uniform(-0.5, 0.5, size=(B, nlev, nx))).astype(np.float32)
    x_main[:, :, 1] = g.uniform(0, 1, size=(B, nlev)).astype(np.float32)
    x_main[:, :, 2] = (1e-4 * g.uniform(0, 1, size=(B, nlev))).astype(np.float32)
    x_main[:, :, 3] = (1e-4 * g.uniform(0, 1, size=(B, nlev))).astype(np.float32)
    x_sfc = (consts["xmean_sca"][None] + consts["xdiv_sca"][None]
             * g.uniform(-0.5, 0.5, size=(B, nx_sfc))).astype(np.float32)
    return x_main, x_sfc


def checksum(*arrays):
    """Order-sensitive uint64 checksum of the raw fp32 bit patterns."""
    acc = np.uint64(1469598103934665603)
    for a in arrays:
        bits = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64).ravel()
        w = (np.arange(bits.size, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(1)) | np.uint64(1)
        with np.errstate(over="ignore"):
            acc = acc * np.uint64(1099511628211) + np.sum(bits * w, dtype=np.uint64)
    return np.array(acc, np.uint64)


# ---- inputs of the data_utils goldens (make_golden_data_utils.py), regenerated from seeds by the tests --------------------------
EVAL_CASES = {"A": (50, 60, 0.0), "B": (7, 0, 0.0), "C": (40, 60, 250.0)}      # tag: (T, L or 0 for scalars, offset)
CRPS_CASES = {"A": (6, 60, 8), "B": (5, 0, 32)}                                # tag: (T, L, S)


def eval_inputs(tag, ncol=384):
    T, L, off = EVAL_CASES[tag]
    r = np.random.Generator(np.random.PCG64(900 + ord(tag)))
    shape = (T, ncol, L) if L else (T, ncol)
    target = (r.standard_normal(shape) * r.uniform(0.1, 3.0, shape[1:]) + off).astype(np.float32)
    pred = (target + r.standard_normal(shape) * 0.3 + 0.05).astype(np.float32)
    return pred, target


def crps_inputs(tag, ncol=384):
    T, L, S = CRPS_CASES[tag]
    r = np.random.Generator(np.random.PCG64(950 + ord(tag)))
    shape = (T, ncol, L) if L else (T, ncol)
    target = r.standard_normal(shape).astype(np.float32)
    sp = (target[..., None] + 0.5 * r.standard_normal(shape + (S,))).astype(np.float32)
    return sp, target


def derived_inputs(hyam, hybm, N=500, nlev=60):
    """(state_t, state_pmid, state_q0001, state_q0002, state_q0003) for the derived-input goldens."""
    r = np.random.Generator(np.random.PCG64(977))
    tair = r.uniform(180.0, 320.0, (N, nlev)).astype(np.float32)
    pmid = (hyam * 1e5 + r.uniform(6e4, 1.05e5, (N, 1)) * hybm).astype(np.float32)
    q1 = (r.uniform(0, 1, (N, nlev)) * 2e-2 * (pmid / 1e5)).astype(np.float32)
    q2 = r.uniform(0, 1e-4, (N, nlev)).astype(np.float32)
    q3 = r.uniform(0, 1e-4, (N, nlev)).astype(np.float32)
    # missing / fill cells: NaN and +-inf temperatures must propagate as they do through np.maximum / np.minimum / the masked sums
    tair[3, 7], tair[4, 8], tair[5, 9] = np.nan, np.inf, -np.inf
    return tair, pmid, q1, q2, q3


# ---- inputs of the generator goldens (make_golden_generator.py), regenerated from seeds by tests/test_generator.py --------------
def generator_chunk(consts, nt=3, nloc=7, seed=5, nx_sfc_in=24):
    """An in-memory stand-in for one HDF5 training file: the four datasets of rnn/utils.py::generator_xy, (ntime, nloc, ...)."""
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


def generator_coeffs(consts, nx, ny, seed=2):
    g = np.random.Generator(np.random.PCG64(seed))
    xm, xd = consts["xmean_lev"], consts["xdiv_lev"].copy()
    xd[xd == 0] = 1.0
    if nx == 16:
        qmean = np.geomspace(2e-6, 8e-3, 60).astype(np.float32)[:, None]
        xm, xd = np.concatenate([xm, qmean], 1), np.concatenate([xd, 4 * qmean], 1)
    ys = (10 ** g.uniform(3, 7, (60, ny))).astype(np.float32)
    return ((xm, xd), (consts["xmean_sca"], consts["xdiv_sca"])), (ys, consts["yscale_sca"])


GENERATOR_VARIANTS = {
    "mp1": dict(mp_mode=1, remove_past_sfc_inputs=True),
    "mp0_qin": dict(mp_mode=0, remove_past_sfc_inputs=True, rh_input_to_q=True, include_q_input=True, output_prune=True),
    "mpm1_rh2q": dict(mp_mode=-1, remove_past_sfc_inputs=True, rh_input_to_q=True, rh_prune=True, qinput_prune=True),
    "mpm2_v5": dict(mp_mode=-2, remove_past_sfc_inputs=True, rh_input_to_q=True, include_q_input=True, v4_to_v5_inputs=True),
    "mp1_sqrt": dict(mp_mode=1, remove_past_sfc_inputs=False, cld_inp_transformation="sqrt", snowhice_fix=False, nx_sfc_in=19),
    "mpm1_v5_prune": dict(mp_mode=-1, remove_past_sfc_inputs=True, v4_to_v5_inputs=True, qinput_prune=True),
    "mp1_renorm": dict(mp_mode=1, remove_past_sfc_inputs=True, renorm=True),        # stored normalised with reference coefficients: undo, re-apply
    "mp1_prev": dict(mp_mode=1, remove_past_sfc_inputs=True, include_prev_inputs=True, include_prev_outputs=True),
}


def generator_setup(consts, lbd_qn, tag):
    """(in-memory datasets, constructor keywords of generator_xy) of one golden variant."""
    kw = dict(GENERATOR_VARIANTS[tag])
    data = generator_chunk(consts, nx_sfc_in=kw.pop("nx_sfc_in", 24))
    nx = 16 if kw.get("include_q_input") else 15
    ny = 5 if kw["mp_mode"] > 0 else 6
    xco, yco = generator_coeffs(consts, nx, ny)
    if kw.get("include_prev_inputs"):
        g = np.random.Generator(np.random.PCG64(3))
        xm = np.concatenate([xco[0][0], g.standard_normal((60, 11)).astype(np.float32)], 1)
        xd = np.concatenate([xco[0][1], (1 + g.random((60, 11))).astype(np.float32)], 1)
        xco = ((xm, xd), xco[1])
    full = dict(xcoeffs=xco, ycoeffs=yco, lbd_qc=consts["lbd_qc"], lbd_qi=consts["lbd_qi"], lbd_qn=lbd_qn,
                hyam=consts["hyam"], hybm=consts["hybm"], **kw)
    if full.pop("renorm", False):
        g = np.random.Generator(np.random.PCG64(11))
        xr = ((g.standard_normal((60, 15)).astype(np.float32) * 0.1, (1 + g.random((60, 15))).astype(np.float32)),
              (g.standard_normal(24).astype(np.float32) * 0.1, (1 + g.random(24)).astype(np.float32)))
        yr = ((10 ** g.uniform(2, 5, (60, 6))).astype(np.float32), (10 ** g.uniform(2, 5, 8)).astype(np.float32))
        # the file holds NORMALISED data: store (x - mean) / div and y * yscale of the physical chunk
        with np.errstate(all="ignore"):
            data["input_lev"] = ((data["input_lev"] - xr[0][0]) / xr[0][1]).astype(np.float32)
            data["input_sca"] = ((data["input_sca"] - xr[1][0]) / xr[1][1]).astype(np.float32)
            data["output_lev"] = (data["output_lev"] * yr[0]).astype(np.float32)
            data["output_sca"] = (data["output_sca"] * yr[1]).astype(np.float32)
        full["xcoeffs_ref"], full["ycoeffs_ref"] = xr, yr
    return data, full
