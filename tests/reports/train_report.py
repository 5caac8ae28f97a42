#!/usr/bin/env python3
"""Prints gradient / loss parity of the HIP training step against the reference gradients and times one
TBPTT optimiser step (GPU box)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from conftest import GOLDEN, load_npz_model, rel_err
from oracle import torch_ref
from climsim_amd.train import Trainer
from synth import synth_inputs

consts, weights, flags = load_npz_model("cur_lstm128")
io = np.load(os.path.join(GOLDEN, "cur_lstm128_io.npz"))
grid = np.load(os.path.join(GOLDEN, "grid_consts.npz"))
ref = torch_ref.EmulatorRef(consts, weights, legacy=False, use_lstm=True, output_prune=bool(flags["output_prune"]), scrub_inf=True)
B, Tw = int(io["grad.B"]), int(io["grad.T_w"])
xr = [torch.from_numpy(io[f"grad.t{t}.x_main"]) for t in range(Tw)]
xs = [torch.from_numpy(io[f"grad.t{t}.x_sfc"]) for t in range(Tw)]
with torch.no_grad():
    pre = [ref.preprocess(a, b) for a, b in zip(xr, xs)]
    tgt, tgt_sfc = torch.from_numpy(io["grad.tgt"]), torch.from_numpy(io["grad.tgt_sfc"])
    yto, yto_sfc = ref.postprocess(tgt, tgt_sfc, torch.cat(xr, 0))
d = lambda t: t.contiguous().cuda()
tr = Trainer(consts, weights, grid["hyai"], grid["hybi"], output_prune=bool(flags["output_prune"]), max_batch=8, max_window=3)
sl = lambda a: [d(a[t * B:(t + 1) * B]) for t in range(Tw)]
sc, mem, dm = tr.window_step([d(p[0]) for p in pre], [d(p[1]) for p in pre], [d(a) for a in xr], sl(tgt), sl(tgt_sfc),
                             sl(yto), sl(yto_sfc), d(torch.from_numpy(io["grad.mem0"])), optimise=False)
print("loss scalars (hip vs reference):")
for k, v in sc.items():
    print(f"  {k:15s} {v:.9e}  {float(io['grad.loss.' + k]):.9e}")
print(f"d_mem0 rel err {rel_err(dm.cpu().numpy(), io['grad.d_mem0']):.2e}")
for n, g in tr.grad_dict().items():
    r = io["grad.dw." + n]
    print(f"  grad {n:28s} rel err {rel_err(g.cpu().numpy().reshape(r.shape), r):.2e}   max|g| {np.abs(r).max():.3e}")

# ---- timing: config 3 shape, 384 columns, T_w = 3 ---------------------------------------------------------
B, Tw = 384, 3
tr = Trainer(consts, weights, grid["hyai"], grid["hybi"], output_prune=True, max_batch=B, max_window=Tw)
g = torch.Generator().manual_seed(0)
xm, xsf = synth_inputs(consts, B, 1)
with torch.no_grad():
    xn, xsn = ref.preprocess(torch.from_numpy(xm), torch.from_numpy(xsf))
    t5 = torch.randn(B, 60, 5, generator=g); t8 = torch.randn(B, 8, generator=g)
    y6, y8 = ref.postprocess(t5, t8, torch.from_numpy(xm))
L = lambda a: [d(a)] * Tw
mem = torch.zeros(60, B, 16, device="cuda")
args = (L(xn), L(xsn), L(torch.from_numpy(xm)), L(t5), L(t8), L(y6), L(y8))
for _ in range(3):
    sc, mem, _ = tr.window_step(*args, mem)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n):
    sc, mem, _ = tr.window_step(*args, mem)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"train step B={B} T_w={Tw}: {dt*1e3:.3f} ms  -> {B*Tw/dt:.0f} column-timesteps/s   loss {sc['loss']:.4f}")
