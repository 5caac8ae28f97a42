"""21,600 columns (the whole high-res grid) in one call on one GPU: finite, and rows agree with a 2,700-column call."""
import sys, time, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/tests/golden"]
import climsim_amd
from conftest import block_errors
from synth import synth_inputs
d = np.load("/root/repo/tests/golden/v4_memory_model.npz")
consts = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
weights = {k[2:]: d[k] for k in d.files if k.startswith("w.")}
B = 21600
m = climsim_amd.NewModel_constraint(consts, weights, max_batch=B)
xm, xs = synth_inputs(consts, B, 3)
g = np.random.Generator(np.random.PCG64(1))
args = [torch.from_numpy(a).cuda() for a in (xm, xs, (0.3 * g.standard_normal((B, 60, 16))).astype(np.float32),
                                             g.standard_normal((B, 128)).astype(np.float32), g.standard_normal((B, 128)).astype(np.float32))]
y = m.emulator.forward_packed(*args)
torch.cuda.synchronize()
assert torch.isfinite(y).all()
sub = slice(10000, 12700)
ys = m.emulator.forward_packed(*[a[sub].contiguous() for a in args])
err = block_errors(y[sub].cpu().numpy(), ys.cpu().numpy())
print("21600-column call: finite; rows 10000:12700 vs a 2700-column call:", {k: f"{v:.1e}" for k, v in err.items()})
for _ in range(3):
    m.emulator.forward_packed(*args)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    m.emulator.forward_packed(*args)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"B=21600: {dt * 1e3:.2f} ms per step = {B / dt / 1e6:.2f} M columns/s on one GPU (latency per global high-res time step)")
