import sys, os, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden"): sys.path.insert(0, p)
from test_physrnn_frozen import _load
from make_golden_frozen import inputs_wrapped, draws
from climsim_amd.physrnn import physical_RNN_wrapped
from climsim_amd import _lib
L = ctypes.CDLL(_lib.lib()._name)
g, P, FL = _load("frozen_a153783c")
for B in (48, 384):
    m = physical_RNN_wrapped(P, FL, max_batch=B)
    x, s, mem = (t.cuda() for t in inputs_wrapped(P, B, 5))
    dr = {k: v.cuda() for k, v in draws(FL, B, 6).items()}
    for _ in range(5): m(x, s, mem, hx2=dr["hx2"], mask_u=dr["mask_u"])
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.csa_phys_debug_phase_cycles(buf, 1)
    N = 50
    for _ in range(N): m(x, s, mem, hx2=dr["hx2"], mask_u=dr["mask_u"])
    torch.cuda.synchronize()
    L.csa_phys_debug_phase_cycles(buf, 0)
    v = np.array(list(buf), float) / (N * B)
    print("B", B, "cycles per workgroup by phase A,B,C,D,E(levels),F(cloud):", [int(t) for t in v[:6]], "sum", int(v[:6].sum()), "(100 MHz counter?)")
