"""Step latency with and without hipGraph replay (memory wrapper, persistent buffers, non-default stream)."""
import sys, time, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests/golden"]
import climsim_amd
from synth import synth_inputs
d = np.load("/root/repo/tests/golden/v4_memory_model.npz")
consts = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
weights = {k[2:]: d[k] for k in d.files if k.startswith("w.")}
st = torch.cuda.Stream()
for B in (1, 48, 256, 384):
    m = climsim_amd.NewModel_constraint(consts, weights, max_batch=B)
    xm, xs = synth_inputs(consts, B, 1)
    dv = lambda a: torch.from_numpy(a).cuda()
    args = (dv(xm), dv(xs), torch.zeros(B, 60, 16, device="cuda"), torch.randn(B, 128, device="cuda"), torch.randn(B, 128, device="cuda"))
    out = torch.empty(B, m.emulator.packed_width, device="cuda")
    torch.cuda.synchronize()
    res = {}
    with torch.cuda.stream(st):
        for mode in (0, 1, 0, 1):
            m.emulator.set_graph(bool(mode))
            for _ in range(20):
                m.emulator.forward_packed(*args, out=out)
            st.synchronize(); t0 = time.perf_counter()
            for _ in range(300):
                m.emulator.forward_packed(*args, out=out)
            st.synchronize()
            res.setdefault(mode, []).append((time.perf_counter() - t0) / 300 * 1e6)
        m.emulator.set_graph(False)
    print(f"B={B:4d}  launches {min(res[0]):7.1f} us   graph replay {min(res[1]):7.1f} us   ratio {min(res[0]) / min(res[1]):.2f}")
