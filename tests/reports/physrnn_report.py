#!/usr/bin/env python3
"""Measured error of the HIP physRNN path against every shipped-artefact fixture, next to the artefact's own float32 noise
(artefact vs the float64 restatement).  Writes profiles/<round>_physrnn_parity.txt when run on the GPU box:
    python tests/reports/physrnn_report.py > gpurun_out/physrnn_parity.txt"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from make_golden_physrnn import inputs, inputs_rad
from oracle import physrnn_ref, physrnn_rad_ref
from climsim_amd.physrnn import physical_RNN_autoreg

FIX = [("physrnn_hidden", "num14564_BEST", 2), ("physrnn_hidden_ep40", "num14564_ep40", 1), ("physrnn_hidden_b", "num49672_BEST", 1),
       ("physrnn_rad", "num4050_BEST", 2), ("physrnn_rad_nomcica", "num71535_BEST", 1), ("physrnn_rad_liqfrac", "num83000_ep20", 1),
       ("physrnn_rad_stoch_a", "num5730_BEST", 1), ("physrnn_rad_stoch_b", "num62104_BEST", 1), ("physrnn_rad_stoch_c", "num62104_BEST_ep11", 1),
       ("physrad16_a", "physRad-16_nreg16 num14751_BEST", 1), ("physrad16_b", "physRad-16_nreg16 num55617_BEST", 1),
       ("physrad16_c", "physRad-16_nreg16 num55617_ep12", 1), ("physrad16_nh96", "physRad-16_nreg16 neur96 num20600_BEST", 1),
       ("physrad16_nh112_a", "physRad-16_nreg16 neur112 num34341_BEST", 1), ("physrad16_nh112_b", "physRad-16_nreg16 neur112 num37201_BEST", 1),
       ("physrad4_a", "physRad-16_nreg4 neur112 num35741_BEST", 1), ("physrad4_b", "physRad-16_nreg4 neur112 num95220_BEST", 1),
       ("physrad16_nh112_cld", "physRad-16_nreg16 neur112 num88955_BEST", 1),
       ("physrad16_e3sm", "physRad-16_nreg16 neur128 num94634_BEST (physics_rad_e3sm)", 1),
       ("physrad16_e3sm_cld", "physRad-16_nreg16 neur112 num88741_BEST (physics_rad_e3sm, learned cloud optics)", 1),
       ("physrad16_b_gpu", "physRad-16_nreg16 num55617_BEST, the _script_gpu twin (another checkpoint)", 1)]
print("fixture | artefact | case B | block: HIP-vs-artefact max|err| / max|ref| ; artefact-vs-float64 restatement (its own rounding) / max|ref|")
for name, tag, ncase in FIX:
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    P = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    rad = "gas_optics_model_lw.mlp1.weight" in P
    m = physical_RNN_autoreg(P, max_batch=64)
    for i in range(ncase):
        B, seed = (int(v) for v in g[f"case{i}.cfg"])
        xm, xs, mem, xd = (inputs_rad if rad else inputs)(P, B, seed)
        hx2 = torch.from_numpy(g[f"case{i}.hx2"])
        nz = {k: torch.from_numpy(g[f"case{i}.{k}"]) for k in ("hx1", "eps3", "srnn") if f"case{i}.{k}" in g.files}
        hip_kw = {"_srnn": nz["srnn"].cuda()} if "srnn" in nz else {k: v.cuda() for k, v in nz.items()}   # physRad: teacher-forced rnn3
        lm = getattr(m, "physrad", False)                       # that family takes / returns rnn_mem level-major
        got = [t.cpu().double() for t in m([xm.cuda(), xs.cuda(), (mem.transpose(0, 1).contiguous() if lm else mem).cuda(), xd.cuda()],
                                           hx2=hx2.cuda(), **hip_kw)]
        if lm:
            got[2] = got[2].transpose(0, 1)
        ref = [torch.from_numpy(g[f"case{i}.{k}"]).double() for k in ("out", "out_sfc", "mem_out")]
        P64 = {k: v.double() for k, v in P.items()}
        fwd = physrnn_rad_ref.forward if rad else physrnn_ref.forward
        r64 = fwd(P64, xm.double(), xs.double(), mem.double(), xd.double(), hx2.double(), **{k: v.double() for k, v in nz.items()})
        cells = []
        for bn, a, b, c in (("out", got[0], ref[0], r64[0]), ("out_sfc", got[1], ref[1], r64[1])):
            for col in range(a.shape[-1]):
                sc = b[..., col].abs().max().item()
                if sc > 0:
                    cells.append(f"{bn}[{col}] {(a[..., col] - b[..., col]).abs().max().item() / sc:.1e};{(b[..., col] - c[..., col]).abs().max().item() / sc:.1e}")
        sc = ref[2].abs().max().item()
        cells.append(f"mem {(got[2] - ref[2]).abs().max().item() / sc:.1e};{(ref[2] - r64[2]).abs().max().item() / sc:.1e}")
        print(f"{name} | {tag} | case {i} B={B} | " + "  ".join(cells))
