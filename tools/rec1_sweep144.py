"""nh = 144 tuple wrapper: one-column kernel up to various batch thresholds vs the two-column kernel (development tool)."""
import sys, time, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests/golden"]
import climsim_amd
from synth import synth_inputs
for tag in ("cur_lstm144", "cur_lstm128"):
    d = np.load(f"/root/repo/tests/golden/{tag}_model.npz")
    consts = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
    weights = {k[2:]: d[k] for k in d.files if k.startswith("w.")}
    for B in (256, 384, 512, 768):
        m = climsim_amd.model_wrapper(consts, weights, use_lstm=True, output_prune=True, max_batch=B)
        xm, xs = synth_inputs(consts, B, 1)
        args = (torch.from_numpy(xm).cuda(), torch.from_numpy(xs).cuda(), torch.zeros(60, B, 16, device="cuda"))
        res = {}
        for thr in (0, 4096, 0, 4096):
            m.emulator.set_rec1_max_batch(thr)
            m.emulator.set_halves(False)
            for _ in range(10):
                m(*args)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100):
                m(*args)
            torch.cuda.synchronize()
            res.setdefault(thr, []).append((time.perf_counter() - t0) / 100 * 1e6)
        print(f"{tag} B={B:4d}  two-column {min(res[0]):7.1f} us   one-column {min(res[4096]):7.1f} us")
