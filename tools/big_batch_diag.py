"""Diagnostic: rows of a large call vs the same rows computed in a small call, for several sizes / settings."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import climsim_amd
from conftest import block_errors
from synth import synth_inputs
d = np.load(os.path.join(ROOT, "tests", "golden", "v4_memory_model.npz"))
consts = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
weights = {k[2:]: d[k] for k in d.files if k.startswith("w.")}
BM = 21600
m = climsim_amd.NewModel_constraint(consts, weights, max_batch=BM)
xm, xs = synth_inputs(consts, BM, 3)
g = np.random.Generator(np.random.PCG64(1))
full = [torch.from_numpy(a).cuda() for a in (xm, xs, (0.3 * g.standard_normal((BM, 60, 16))).astype(np.float32),
                                             g.standard_normal((BM, 128)).astype(np.float32), g.standard_normal((BM, 128)).astype(np.float32))]
for halves in (True, True, True):
    m.emulator.set_halves(halves)
    for B in (2700, 5400, 8192, 10800, 16384, 21600):
        args = [a[:B].contiguous() for a in full]
        y = m.emulator.forward_packed(*args)
        worst = {}
        for lo in range(0, B, 700):
            hi = min(B, lo + 700)
            ys = m.emulator.forward_packed(*[a[lo:hi].contiguous() for a in args])
            e = block_errors(y[lo:hi].cpu().numpy(), ys.cpu().numpy())
            for k, v in e.items():
                if v > worst.get(k, (0, 0))[0]:
                    worst[k] = (v, lo)
        print(f"halves={halves} B={B}: worst block error vs 700-column calls", {k: (f"{v[0]:.1e}", v[1]) for k, v in worst.items()}, flush=True)
m.emulator.set_halves(None)
