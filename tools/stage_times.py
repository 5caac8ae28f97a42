"""Per-stage HIP-event times (csa_set_profiling) of the tuple wrapper for a few current-generation models (development tool)."""
import sys, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/tests/golden"]
import climsim_amd
from synth import synth_inputs
for tag, B in (("cur_lstm128", 384), ("cur_lstm144", 384), ("cur_lstm144", 256), ("cur_gru128", 384), ("cur_lstm128", 48), ("cur_gru128", 48), ("cur_gru128", 256)):
    d = np.load(f"/root/repo/tests/golden/{tag}_model.npz")
    consts = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
    weights = {k[2:]: d[k] for k in d.files if k.startswith("w.")}
    m = climsim_amd.model_wrapper(consts, weights, use_lstm="lstm" in tag, output_prune="lstm" in tag, max_batch=B)
    xm, xs = synth_inputs(consts, B, 1)
    args = (torch.from_numpy(xm).cuda(), torch.from_numpy(xs).cuda(), torch.zeros(60, B, 16, device="cuda"))
    for _ in range(10):
        m(*args)
    m.emulator.set_profiling(True); m.emulator.reset_profile()
    for _ in range(50):
        m(*args)
    prof, n = m.emulator.get_profile()
    m.emulator.set_profiling(False)
    print(tag, B, {k: round(1e3 * v, 1) for k, v in prof.items()}, "sum", round(1e3 * sum(prof.values()), 1), "us")
