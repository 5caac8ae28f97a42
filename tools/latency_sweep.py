"""Step latency of the packed memory wrapper at small batches: one-column (default up to 256) vs two-column recurrent kernel."""
import sys, time, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests/golden"]
import climsim_amd
from synth import synth_inputs
d = np.load("/root/repo/tests/golden/v4_memory_model.npz")
consts = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
weights = {k[2:]: d[k] for k in d.files if k.startswith("w.")}
for B in (1, 16, 48, 128, 192, 256, 320, 384):
    m = climsim_amd.NewModel_constraint(consts, weights, max_batch=B)
    xm, xs = synth_inputs(consts, B, 1)
    dv = lambda a: torch.from_numpy(a).cuda()
    args = (dv(xm), dv(xs), torch.zeros(B, 60, 16, device="cuda"), torch.randn(B, 128, device="cuda"), torch.randn(B, 128, device="cuda"))
    out = torch.empty(B, m.emulator.packed_width, device="cuda")
    res = {}
    for mode in (256, 0, 256, 0):
        m.emulator.set_rec1_max_batch(mode)
        for _ in range(20):
            m.emulator.forward_packed(*args, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300):
            m.emulator.forward_packed(*args, out=out)
        torch.cuda.synchronize()
        res.setdefault(mode, []).append((time.perf_counter() - t0) / 300 * 1e6)
    print(f"B={B:4d}  one-column (<=256) {min(res[256]):7.1f} us   two-column {min(res[0]):7.1f} us   ratio {min(res[0]) / min(res[256]):.2f}")
