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
    return tair, pmid, q1, q2, q3
