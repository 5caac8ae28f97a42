"""Diagnostic: large two-stream calls of the current-generation GRU / LSTM tuple wrapper and of physRNN vs the same rows in small calls."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import climsim_amd
from conftest import load_npz_model
from synth import synth_inputs
for tag in ("cur_gru128", "cur_lstm128"):
    consts, weights, flags = load_npz_model(tag)
    B = 8192
    wrap = climsim_amd.model_wrapper(consts, weights, max_batch=B, use_lstm=bool(flags["use_lstm"]), output_prune=bool(flags["output_prune"]))
    xm, xs = synth_inputs(consts, B, 31)
    mem = (0.3 * np.random.Generator(np.random.PCG64(9)).standard_normal((60, B, 16))).astype(np.float32)
    xm, xs, mem = (torch.from_numpy(a).cuda() for a in (xm, xs, mem))
    wrap.emulator.set_halves(False)
    ref = [wrap(xm[lo:lo + 512].contiguous(), xs[lo:lo + 512].contiguous(), mem[:, lo:lo + 512].contiguous()) for lo in range(0, B, 512)]
    ref = [torch.cat([r[0] for r in ref]), torch.cat([r[1] for r in ref]), torch.cat([r[2] for r in ref], dim=1)]
    for halves in (True, False):
        wrap.emulator.set_halves(halves)
        bad = 0
        for rep in range(6):
            out = wrap(xm, xs, mem)
            bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(out, ref))
        print(tag, "halves", halves, "mismatching tensors in 6 calls:", bad, flush=True)
    wrap.emulator.set_halves(None)
# physRNN (GRU two-column kernel above 256 columns), single stream by construction
from make_golden_physrnn import inputs
from climsim_amd.physrnn import physical_RNN_autoreg
g = np.load(os.path.join(ROOT, "tests", "golden", "physrnn_hidden.npz"))
P = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
B = 4096
m = physical_RNN_autoreg(P, max_batch=B)
xs_ = [t.cuda() for t in inputs(P, B, 5)]
hx2 = torch.randn(B, 128, generator=torch.Generator().manual_seed(1)).cuda()
ref = [m([t[lo:lo + 512].contiguous() for t in xs_], hx2=hx2[lo:lo + 512].contiguous()) for lo in range(0, B, 512)]
ref = [torch.cat([r[k] for r in ref]) for k in range(3)]
bad = 0
for rep in range(6):
    out = m(xs_, hx2=hx2)
    bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(out, ref))
print("physRNN 4096 vs 512-column calls, mismatching tensors in 6 calls:", bad)
