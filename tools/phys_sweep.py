"""physRNN-Hidden step time vs batch size (host floor at small batches). Run on the GPU box: python tools/phys_sweep.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from make_golden_physrnn import inputs
from climsim_amd.physrnn import physical_RNN_autoreg
g = np.load(os.path.join(ROOT, "tests", "golden", "physrnn_hidden.npz"))
P = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
m = physical_RNN_autoreg(P, max_batch=2700)
for B in (1, 8, 48, 256, 384, 768, 2700):
    xs = [t.cuda() for t in inputs(P, B, B)]
    hx2 = torch.randn(B, 128, device="cuda")
    mem = xs[2]
    for _ in range(20):
        _, _, mem = m([xs[0], xs[1], mem, xs[3]], hx2=hx2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        _, _, mem = m([xs[0], xs[1], mem, xs[3]], hx2=hx2)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:5d}  {dt*1e6:8.1f} us/step  {B/dt/1e6:6.3f} M columns/s   (host enqueue {t_host/n*1e6:.1f} us/step)", flush=True)
# long autoregressive feedback on synthetic inputs: does the carried state stay finite, and does the step time change?
for seed in (300, 384):
    B = 384
    xs = [t.cuda() for t in inputs(P, B, seed)]
    hx2 = torch.randn(B, 128, device="cuda")
    mem = xs[2]
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            out, _, mem = m([xs[0], xs[1], mem, xs[3]], hx2=hx2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        print(f"seed {seed} steps {100*(it+1)}: {dt*1e6:.1f} us/step  finite={bool(torch.isfinite(mem).all())} max|mem|={mem.abs().max().item():.3g} max|out|={out.abs().max().item():.3g}", flush=True)
