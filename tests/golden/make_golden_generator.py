#!/usr/bin/env python3
"""Golden vectors from the reference's own training-data generator rnn/utils.py::generator_xy (:1870-2384): its constructor and its
`__getitem__` run in the build container on in-memory datasets, for every target construction (mp_mode 0 / 1 / -1 / -2), the
float64 RH -> q conversion (`relative_to_specific_humidity_climsim`, :647-690), `v4_to_v5_inputs`, the numba normalisers
(:1803-1870), the re-normalisation path (`xcoeffs_ref` / `ycoeffs_ref`) and `include_prev_inputs` / `include_prev_outputs`.

Import notes (as make_golden_wrapper.py): rnn/utils.py names h5py, numba, torchmetrics, torchinfo and matplotlib at module top; none
is installed.  They are bound to inert placeholders; two of them are touched by generator_xy and get the minimum that lets the
reference's OWN code run, nothing of the libraries' behaviour is re-implemented:
  * numba.njit is the identity decorator, so the "numba kernels" run as the plain Python loops they are written as;
  * h5py.File(path, 'r') returns the dict of numpy arrays registered for `path` (+ a no-op close()): the class indexes its datasets
    with `hdf[name][indices, :]` and reads `.shape`, which numpy arrays provide.
Upstream defect recorded here: with `v4_to_v5_inputs` and the exponential cloud transform the numba branch calls
`v4_to_v5_inputs_numba(x_lev_b, liq_frac)` with two arguments where the function takes three (:1803, :2216) -- a TypeError on every
call.  The script asserts that it raises and pins the class's non-numba branch (`use_numba = False`, :2220-2230) for those variants.

Inputs are regenerated from seeds by tests/golden/synth.py (shared with tests/test_generator.py); only outputs are stored
(generator_golden.npz, data only)."""
import os
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from synth import GENERATOR_VARIANTS, generator_setup  # noqa: E402
from make_golden_wrapper import import_wrapper  # noqa: E402
from make_golden_current import consts  # noqa: E402

NAMES = ["x_lev", "x_sfc", "y_lev", "y_sfc", "x_lev_denorm", "y_lev_denorm", "y_sfc_denorm"]
_FILES = {}


class _MemFile(dict):
    def __init__(self, path, mode="r"):
        super().__init__(_FILES[path])

    def close(self):
        pass


def main():
    ref_models, ref_utils = import_wrapper()
    sys.modules["h5py"].File = _MemFile
    ref_utils.h5py.File = _MemFile
    c = consts()
    out = {}
    for tag in GENERATOR_VARIANTS:
        data, full = generator_setup(c, c["lbd_qn"], tag)
        _FILES[tag] = {k: v.copy() for k, v in data.items()}
        gen = ref_utils.generator_xy(tag, nloc=7, **full)
        prev = full.get("include_prev_inputs") or full.get("include_prev_outputs")
        idx = [1, 2] if prev else [0, 2]
        with np.errstate(all="ignore"):
            if full.get("v4_to_v5_inputs") and full.get("cld_inp_transformation", "exp") == "exp":
                try:
                    gen[list(idx)]
                    raise SystemExit("expected the numba v4_to_v5 branch to raise (rnn/utils.py:2216)")
                except TypeError as e:
                    print(tag, "numba branch raises upstream:", e)
                _FILES[tag] = {k: v.copy() for k, v in data.items()}
                gen.use_numba = False
            res = gen[list(idx)]
        assert len(gen) == (2 if prev else 3) * 7
        for name, a in zip(NAMES, res):
            a = a.numpy() if hasattr(a, "numpy") else np.asarray(a)
            out[f"{tag}.{name}"] = a.astype(np.float32) if a.dtype != np.float32 else a
            print(tag, name, a.shape, a.dtype)
        out[f"{tag}.idx"] = np.array(idx, np.int32)
        out[f"{tag}.dims"] = np.array([gen.nx, gen.nx_sfc, gen.ny, gen.ny_sfc], np.int32)
        if prev:        # first index 0 must raise
            try:
                gen[[0, 1]]
                raise SystemExit("expected NotImplementedError")
            except NotImplementedError:
                pass
    # the float64 humidity conversion on its own (rnn/utils.py:675-690): a temperature sweep across all three branches of eice
    T = np.linspace(150.0, 320.0, 341)
    rh = np.linspace(0.0, 1.2, 341)[::-1].copy()
    p = np.geomspace(10.0, 101325.0, 341)
    out["rh2q.T"], out["rh2q.rh"], out["rh2q.p"] = T, rh, p
    out["rh2q.q"] = np.asarray(ref_utils.relative_to_specific_humidity_climsim(rh, T, p), np.float64)
    out["rh2q.eliq"], out["rh2q.eice"] = np.asarray(ref_utils.eliq(T), np.float64), np.asarray(ref_utils.eice(T), np.float64)
    np.savez_compressed(os.path.join(OUT, "generator_golden.npz"), **out)
    print("wrote generator_golden.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    if not os.path.isdir("/root/reference/rnn"):
        sys.exit("reference not present: golden fixtures can only be regenerated in the build container")
    main()
