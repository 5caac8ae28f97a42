"""Throughput of the device-side training-data generator (csrc/gen.hip) on a chunk already in HBM.
Run on the GPU box: python tools/gen_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
from test_generator import _setup, VARIANTS
from climsim_amd.generator import generator_xy

for vi in (0, 3):
    consts, data, full = _setup(VARIANTS[vi])
    gen = generator_xy(data, nloc=data["input_lev"].shape[1], **full)
    c = gen.cfg
    for N in (384 * 8, 384 * 64):
        g = torch.Generator().manual_seed(N)
        rep = lambda a: torch.from_numpy(np.ascontiguousarray(a.reshape((-1,) + a.shape[2:]), np.float32))
        def tile(a):
            t = rep(a)
            k = (N + t.shape[0] - 1) // t.shape[0]
            return t.repeat((k,) + (1,) * (t.dim() - 1))[:N].contiguous().cuda()
        xl, xs, yl, ys = tile(data["input_lev"]), tile(data["input_sca"]), tile(data["output_lev"]), tile(data["output_sca"])
        for _ in range(3):
            out = gen.prepare(xl, xs, yl, ys)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            out = gen.prepare(xl, xs, yl, ys)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        nbytes = sum(t.numel() for t in (xl, xs, yl, ys)) * 4 + sum(t.numel() for t in out) * 4
        print(f"variant {vi} (mp_mode {VARIANTS[vi]['mp_mode']}), N = {N} columns: {dt*1e6:.1f} us per chunk, "
              f"{nbytes/1e6:.0f} MB in+out -> {nbytes/dt/1e12:.2f} TB/s, {N/dt/1e6:.1f} M columns/s", flush=True)
