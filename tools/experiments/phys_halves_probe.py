#!/usr/bin/env python3
"""Experiment (round 3): would the deployed physRNN export gain from running two column halves on two streams, as the LSTM
path does from 640 columns?  Two handles of B/2 columns on two torch streams against one handle of B columns; no library change."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from test_physrnn_frozen import _load
from make_golden_frozen import inputs_wrapped, draws
from climsim_amd.physrnn import physical_RNN_wrapped

g, P, FL = _load("frozen_a153783c")
for B in (384, 768, 2700):
    H = B // 2
    whole = physical_RNN_wrapped(P, FL, max_batch=B)
    halves = [physical_RNN_wrapped(P, FL, max_batch=H) for _ in range(2)]
    x, s, mem = (t.cuda() for t in inputs_wrapped(P, B, 5))
    dr = {k: v.cuda() for k, v in draws(FL, B, 6).items()}
    xs = [(x[i * H:(i + 1) * H].contiguous(), s[i * H:(i + 1) * H].contiguous(), mem[:, i * H:(i + 1) * H].contiguous(),
           dr["hx2"][i * H:(i + 1) * H].contiguous(), dr["mask_u"][:, i * H:(i + 1) * H].contiguous()) for i in range(2)]
    st = [torch.cuda.Stream() for _ in range(2)]

    def one():
        whole(x, s, mem, hx2=dr["hx2"], mask_u=dr["mask_u"])

    def two():
        cur = torch.cuda.current_stream()
        for i in range(2):
            st[i].wait_stream(cur)
            with torch.cuda.stream(st[i]):
                halves[i](xs[i][0], xs[i][1], xs[i][2], hx2=xs[i][3], mask_u=xs[i][4])
        for i in range(2):
            cur.wait_stream(st[i])

    for name, f in (("one call", one), ("two halves", two)):
        for _ in range(30):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            f()
        torch.cuda.synchronize()
        print(B, name, round((time.perf_counter() - t0) / 300 * 1e3, 4), "ms", flush=True)
