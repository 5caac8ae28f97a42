#!/usr/bin/env python3
"""Prints the per-block parity errors of the HIP path against every golden vector set (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from conftest import GOLDEN, block_errors, conditioning_tol, load_npz_model, rel_err
from synth import synth_inputs
import climsim_amd

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
print("device:", torch.cuda.get_device_name(0))
consts, weights, _ = load_npz_model("v4_stateless")
io = np.load(os.path.join(GOLDEN, "v4_stateless_io.npz"))
m = climsim_amd.NewModel_constraint(consts, weights, max_batch=512)
print(f"stateless v4 wrapper (artefact is fp32-ill-conditioned; tolerance {conditioning_tol('v4_stateless'):.2e}):")
for B in (1, 8, 67, 384):
    xm, xs = (io[f"B{B}.x_main"], io[f"B{B}.x_sfc"]) if f"B{B}.x_main" in io.files else synth_inputs(consts, B, int(io[f"B{B}.seed"]))
    y = m(dev(xm), dev(xs), noise=(dev(io[f"B{B}.hx2"]), dev(io[f"B{B}.cx2"]))).cpu().numpy()
    print(f"  B={B:4d}  vs artefact output:", {k: f"{v:.2e}" for k, v in block_errors(y, io[f"B{B}.yout"]).items()})
consts, weights, _ = load_npz_model("v4_memory")
io = np.load(os.path.join(GOLDEN, "v4_memory_io.npz"))
m = climsim_amd.NewModel_constraint(consts, weights, max_batch=512)
print("memory v4 wrapper rollout (tolerance 1e-5 of block max):")
for B in (1, 8, 384):
    mem = torch.zeros(B, 60, 16, device="cuda")
    for t in range(int(io[f"B{B}.nsteps"])):
        p = f"B{B}.t{t}."
        xm, xs = (io[p + "x_main"], io[p + "x_sfc"]) if p + "x_main" in io.files else synth_inputs(consts, B, int(io[p + "seed"]))
        y = m(dev(xm), dev(xs), mem, noise=(dev(io[p + "hx2"]), dev(io[p + "cx2"])))
        print(f"  B={B:4d} t={t}  vs artefact output:", {k: f"{v:.2e}" for k, v in block_errors(y.cpu().numpy(), io[p + "yout"]).items()})
        mem = y[:, 368:].reshape(B, 60, 16).contiguous()
for tag in ("cur_lstm128", "cur_gru128"):
    consts, weights, flags = load_npz_model(tag)
    io = np.load(os.path.join(GOLDEN, f"{tag}_io.npz"))
    kw = dict(use_lstm=bool(flags["use_lstm"]), output_prune=bool(flags["output_prune"]))
    model = climsim_amd.RNN_autoreg(consts, weights, max_batch=64, **kw)
    print(f"{tag} vs reference class RNN_autoreg:")
    for B in (2, 16):
        for t in range(int(io[f"B{B}.nsteps"])):
            p = f"B{B}.t{t}."
            o, os_, mo = model([dev(io[p + "x_main_n"]), dev(io[p + "x_sfc_n"]), dev(io[p + "mem_in"])])
            print(f"  B={B:2d} t={t}  out {rel_err(o.cpu().numpy(), io[p+'out']):.2e}  out_sfc {rel_err(os_.cpu().numpy(), io[p+'out_sfc']):.2e}  mem {rel_err(mo.cpu().numpy(), io[p+'mem_out']):.2e}")
