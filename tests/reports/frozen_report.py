"""GPU report: HIP error of the frozen-export path per output block, in units of the block's float32 rounding level
(profiles/r3_physrnn_frozen_parity.txt).  Run on the GPU box from the repo root."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import test_physrnn_frozen as T
from oracle import physrnn_frozen_ref as R
from oracle import physrnn_ref as D
from climsim_amd.physrnn import physical_RNN_wrapped

for fx in T.FIX:
    g, P, FL = T._load(fx)
    m = physical_RNN_wrapped(P, FL, max_batch=64)
    for i in range(2):
        x, s, mem, dr, ref = T._case(g, P, i)
        d = lambda t: None if t is None else t.cuda()
        got = [t.cpu() for t in m(d(x), d(s), d(mem), hx2=d(dr["hx2"]), hx1=d(dr.get("hx1")), eps3=d(dr.get("eps3")), mask_u=d(dr["mask_u"]), _srnn=d(dr.get("srnn")))]
        got32 = R.forward(P, FL, x, s, mem, dr["hx2"], dr["mask_u"], **{k: dr[k] for k in ("hx1", "eps3", "srnn") if k in dr})
        got64 = T._f64(P, FL, x, s, mem, dr)
        jit = []
        for sd in range(1, 7):
            R._JITTER = D._JITTER = torch.Generator().manual_seed(sd)
            jit.append(T._blocks(*R.forward(P, FL, x, s, mem, dr["hx2"], dr["mask_u"], **{k: dr[k] for k in ("hx1", "eps3", "srnn") if k in dr})))
            R._JITTER = D._JITTER = None
        b_ref, bh, b32, b64 = T._blocks(*ref), T._blocks(*got), T._blocks(*got32), T._blocks(*got64)
        out = []
        for key in b_ref:
            scale = b_ref[key].abs().max().item() + 1e-300
            noise = max((r[key].double() - b64[key]).abs().max().item() for r in [b_ref, b32] + jit)
            e = (bh[key].double() - b64[key]).abs().max().item()
            out.append(f"{key[0][4:]}{key[1] if key[1] is not None else ''}{'t' if len(key) > 2 else ''}: {e / scale:.1e} ({e / max(noise, 1e-5 * scale):.1f})")
        print(fx, "nreg", FL["nreg"], "rnn3", FL["rnn3"], "liq", FL["pred_subgrid_liq_frac"], "case", i, "| error / block max (error / max(noise, 1e-5 max)):", "  ".join(out), flush=True)
