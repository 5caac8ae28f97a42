"""Times the HBM-bound evaluation kernels (csrc/evalm.hip) and the online wrapper's pre/post passes.
Run on the GPU box: python tools/eval_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from climsim_amd.data_utils import data_utils


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


du = data_utils(384)
for T in (100, 1000, 4000):
    p = torch.randn(T, 384, 60, device="cuda")
    t = torch.randn(T, 384, 60, device="cuda")
    dt = timeit(lambda: du.calc_all(p, t, True))
    print(f"eval_metrics T={T}: {dt*1e6:.1f} us  {2*p.numel()*4/dt/1e9:.0f} GB/s algorithmic", flush=True)
for T, S in ((100, 8), (100, 32), (400, 32)):
    sp = torch.randn(T, 384, 60, S, device="cuda")
    t = torch.randn(T, 384, 60, device="cuda")
    dt = timeit(lambda: du.calc_CRPS(sp, t, True))
    print(f"eval_crps T={T} S={S}: {dt*1e6:.1f} us  {(sp.numel()+t.numel())*4/dt/1e9:.0f} GB/s algorithmic", flush=True)
