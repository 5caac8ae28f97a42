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
