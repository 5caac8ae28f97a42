"""Times the packed memory wrapper with the column-half path on/off over a range of batch sizes (development tool)."""
import sys, time, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests/golden"]
import climsim_amd
from synth import synth_inputs
d = np.load("/root/repo/tests/golden/v4_memory_model.npz")
consts = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
weights = {k[2:]: d[k] for k in d.files if k.startswith("w.")}
for B in (192, 384, 768, 1536, 2700, 5400):
    m = climsim_amd.NewModel_constraint(consts, weights, max_batch=B)
    xm, xs = synth_inputs(consts, B, 1)
    dv = lambda a: torch.from_numpy(a).cuda()
    args = (dv(xm), dv(xs), torch.zeros(B, 60, 16, device="cuda"), torch.randn(B, 128, device="cuda"), torch.randn(B, 128, device="cuda"))
    out = torch.empty(B, m.emulator.packed_width, device="cuda")
    res = {}
    for hv in (0, 1, 0, 1):
        m.emulator.set_halves(bool(hv))
        for _ in range(20):
            m.emulator.forward_packed(*args, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 200 if B <= 1536 else 60
        for _ in range(n):
            m.emulator.forward_packed(*args, out=out)
        torch.cuda.synchronize()
        res.setdefault(hv, []).append((time.perf_counter() - t0) / n * 1e6)
    print(f"B={B:5d}  single {min(res[0]):8.1f} us   halves {min(res[1]):8.1f} us   ratio {min(res[0]) / min(res[1]):.3f}")
