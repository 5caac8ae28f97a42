#!/usr/bin/env python3
"""Golden vectors from the reference's own deployment wrapper class rnn/utils.py::model_wrapper (:72-295) around its own
rnn/models/models.py::RNN_autoreg, run in the build container (CPU, torch only).

Import notes.  rnn/utils.py imports, at module top, h5py, numba (njit / config / threading_layer), torchmetrics
(R2Score), torchinfo (summary) and matplotlib -- for its HDF5 generator, its numba data kernels, its evaluation loop and
its plots.  None of them is installed here and NONE is reached by model_wrapper (its methods use torch only).  As for
omegaconf in make_golden_current.py, those names are bound to inert placeholder modules before the import (njit becomes
the identity decorator; nothing else is ever called): nothing of the absent libraries is re-implemented, and a call into
any of them would raise.  No reference source or bytecode is written anywhere: outputs are fp32 arrays in .npz files.

Variants (all on the class's own seeded random initialisation; constants from the shipped artefacts' buffers):
  wrap_v4     nx = 15, v4 inputs, snowhice_fix + qinput_prune + rh_prune on, one 1e10 snow depth, one NaN and one Inf input
  wrap_qin    nx = 16, include_q_input=True  (q from RH, T, p appended as the 16th level input; utils.py:262-270)
  wrap_rh2q   nx = 15, rh_to_q=True          (q replaces RH; utils.py:271-272)
  wrap_v5     nx = 15, v5_input=True          (qn = qliq + qice with lbd_qn, liquid fraction from T; utils.py:186-198)
  wrap_v5p    as wrap_v5 with qinput_prune   (the prune acts on qn BEFORE the transform: utils.py:188-190)
"""
import glob
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/rnn"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from synth import synth_inputs  # noqa: E402
from make_golden_current import consts, import_reference, make_cfg  # noqa: E402

torch.set_num_threads(4)


class _Inert(types.ModuleType):
    """Placeholder for a library rnn/utils.py names at import time but model_wrapper never uses: any attribute is a callable
    that refuses to run (except numba.njit / config, which only decorate / configure)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def refuse(*a, **k):
            raise RuntimeError(f"{self.__name__}.{name} is not available in this container (placeholder)")
        return refuse


def import_wrapper():
    ref_models, ref_metrics = import_reference()            # omegaconf / load_inline neutralised, sys.path set
    for name in ("h5py", "torchmetrics", "torchmetrics.regression", "torchinfo"):
        if name not in sys.modules:
            sys.modules[name] = _Inert(name)
    sys.modules["torchmetrics.regression"].R2Score = object
    nb = _Inert("numba")
    nb.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))      # decorator only
    nb.config = types.SimpleNamespace()
    nb.threading_layer = lambda: "none"
    nb.prange = range
    sys.modules.setdefault("numba", nb)
    try:
        import matplotlib  # noqa: F401
    except Exception:
        sys.modules["matplotlib"] = _Inert("matplotlib")
        sys.modules["matplotlib.pyplot"] = _Inert("matplotlib.pyplot")
    import utils as ref_utils
    return ref_models, ref_utils


def main():
    ref_models, ref_utils = import_wrapper()
    c = consts()
    base = {k: c[k] for k in ("yscale_lev", "yscale_sca", "xmean_lev", "xmean_sca", "xdiv_lev", "xdiv_sca", "hyai", "hybi",
                              "hyam", "hybm", "lbd_qc", "lbd_qi", "lbd_qn")}
    variants = {
        "wrap_v4": dict(nx=15, kw=dict(qinput_prune=True, rh_prune=True, snowhice_fix=True, include_q_input=False)),
        "wrap_qin": dict(nx=16, kw=dict(include_q_input=True, snowhice_fix=False)),
        "wrap_rh2q": dict(nx=15, kw=dict(rh_to_q=True, include_q_input=False, snowhice_fix=False)),
        "wrap_v5": dict(nx=15, kw=dict(v5_input=True, include_q_input=False, snowhice_fix=False)),
        "wrap_v5p": dict(nx=15, kw=dict(v5_input=True, qinput_prune=True, include_q_input=False, snowhice_fix=True)),
    }
    for i, (tag, v) in enumerate(variants.items()):
        torch.manual_seed(500 + i)
        coeffs = dict(base)
        if v["nx"] == 16:     # 16th level input = specific humidity: min-max style coefficients of its own (O(1e-2) kg/kg)
            coeffs["xmean_lev"] = np.concatenate([base["xmean_lev"], np.full((60, 1), 4e-3, np.float32)], 1)
            coeffs["xdiv_lev"] = np.concatenate([base["xdiv_lev"], np.full((60, 1), 2e-2, np.float32)], 1)
        if tag == "wrap_rh2q":   # input 1 is q after the conversion: scale it like q
            coeffs["xmean_lev"] = base["xmean_lev"].copy(); coeffs["xdiv_lev"] = base["xdiv_lev"].copy()
            coeffs["xmean_lev"][:, 1] = 4e-3; coeffs["xdiv_lev"][:, 1] = 2e-2
        cfg = make_cfg(use_lstm=True, nneur=(128, 128), output_prune=True)
        cfg.nx = v["nx"]
        model = ref_models.RNN_autoreg(cfg, coeffs, torch.device("cpu")).eval()
        wrap = ref_utils.model_wrapper(model, mp_mode=1, predict_fluxes=False, **v["kw"]).eval()
        sd = {k: p.detach().numpy().astype(np.float32) for k, p in model.named_parameters()}
        d = {"c." + k: coeffs[k] for k in ("xmean_lev", "xdiv_lev", "xmean_sca", "xdiv_sca", "lbd_qc", "lbd_qi", "lbd_qn",
                                           "yscale_lev", "yscale_sca", "hyam", "hybm")}
        d.update({"w." + k: val for k, val in sd.items()})
        for k, val in v["kw"].items():
            d["flags." + k] = np.array(int(val), np.int32)
        d["flags.use_lstm"] = np.array(1, np.int32); d["flags.output_prune"] = np.array(1, np.int32)
        np.savez(f"{OUT}/{tag}_model.npz", **d)
        io = {}
        for B, seed in ((3, 61), (17, 62)):
            mem = torch.zeros(60, B, 16)
            nsteps = 2
            io[f"B{B}.nsteps"] = np.array(nsteps, np.int32)
            for t in range(nsteps):
                x_main, x_sfc = synth_inputs(c, B, seed * 100 + t + 7 * i)
                if tag in ("wrap_qin", "wrap_rh2q"):
                    # physically plausible humidity: RH scaled by p/ps so that q = RH * qsat(T, p) stays O(1e-2) kg/kg at the
                    # model top too (qsat ~ 1/p; the synthetic uniform RH would give q ~ 1e6 at 10 Pa and the test would
                    # only see those entries)
                    x_main[:, :, 1] *= (c["hyam"] + c["hybm"]).astype(np.float32)[None, :]
                    x_main[:, :, 0] = np.clip(x_main[:, :, 0], 150.0, 330.0)      # all three branches of e_ice (T > 273.15, > 185, colder)
                if tag in ("wrap_v4", "wrap_v5p"):          # the special values the wrapper handles
                    x_sfc[0, 15 if x_sfc.shape[1] > 15 else -1] = 3e10          # "snowhice" missing-value marker
                if tag == "wrap_v4":
                    x_main[B - 1, 7, 4] = np.nan
                    x_main[B - 1, 9, 5] = np.inf
                xm, xs = torch.from_numpy(x_main.copy()), torch.from_numpy(x_sfc.copy())
                with torch.no_grad():
                    # the class normalises a clone, but rh_to_q writes q into the CALLER's x_main0 (utils.py:272): pass copies
                    o6, osd, mem_out = wrap(xm.clone(), xs.clone(), mem.clone())
                    xq = xm.clone()
                    if v["kw"].get("rh_to_q") or v["kw"].get("include_q_input"):
                        pres = torch.squeeze(model.hyam * 100000.0 + xs[:, 0:1] * model.hybm)
                        q = wrap.relative_to_specific_humidity_torch(xm[:, :, 1], xm[:, :, 0], pres)
                        xq = torch.cat((xm, q.unsqueeze(2)), 2) if v["kw"].get("include_q_input") else xq
                        if not v["kw"].get("include_q_input"):
                            xq[:, :, 1] = q
                        io[f"B{B}.t{t}.q"] = q.numpy().copy()
                    xn, xsn = wrap.preprocessing(xq.clone(), xs.clone())
                p = f"B{B}.t{t}."
                io[p + "x_main"] = x_main; io[p + "x_sfc"] = x_sfc
                io[p + "x_main_n"] = xn.numpy().copy(); io[p + "x_sfc_n"] = xsn.numpy().copy()
                io[p + "mem_in"] = mem.numpy().copy()
                io[p + "out_lev"] = o6.numpy().copy(); io[p + "out_sfc"] = osd.numpy().copy()
                io[p + "mem_out"] = mem_out.numpy().copy()
                mem = mem_out.detach().clone()
                print(tag, B, t, "finite", bool(torch.isfinite(o6).all()), float(o6.abs().max()), float(xn.abs().max()))
        np.savez_compressed(f"{OUT}/{tag}_io.npz", **io)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present: golden fixtures can only be regenerated in the build container")
    main()
