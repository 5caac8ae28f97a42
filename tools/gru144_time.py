"""Step time of the current-generation GRU tuple wrapper at nh = 144 (random weights), 384 columns. python tools/gru144_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import climsim_amd
from conftest import load_npz_model
from synth import synth_inputs
consts, _, _ = load_npz_model("cur_gru128")
g = np.random.Generator(np.random.PCG64(144))
nh = 144
shapes = {"mlp_toa1.weight": (nh, 2), "mlp_toa1.bias": (nh,), "mlp_initial.weight": (nh, 16), "mlp_initial.bias": (nh,),
          "mlp_surface1.weight": (nh, 19), "mlp_surface1.bias": (nh,),
          "rnn1.weight_ih_l0": (3 * nh, nh + 16), "rnn1.weight_hh_l0": (3 * nh, nh), "rnn1.bias_ih_l0": (3 * nh,), "rnn1.bias_hh_l0": (3 * nh,),
          "rnn2.weight_ih_l0": (3 * nh, nh), "rnn2.weight_hh_l0": (3 * nh, nh), "rnn2.bias_ih_l0": (3 * nh,), "rnn2.bias_hh_l0": (3 * nh,),
          "mlp_latent.weight": (16, nh), "mlp_latent.bias": (16,), "mlp_output.weight": (5, 16), "mlp_output.bias": (5,),
          "mlp_surface_output.weight": (8, nh), "mlp_surface_output.bias": (8,)}
weights = {k: (g.uniform(-1, 1, s) / np.sqrt(s[-1] if len(s) > 1 else nh)).astype(np.float32) for k, s in shapes.items()}
B = 384
wrap = climsim_amd.model_wrapper(consts, weights, use_lstm=False, max_batch=B, snowhice_fix=True)
xm, xs = synth_inputs(consts, B, 9)
xm, xs = torch.from_numpy(xm).cuda(), torch.from_numpy(xs).cuda()
mem = torch.zeros(60, B, 16, device="cuda")
for _ in range(50):
    _, _, mem = wrap(xm, xs, mem)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    _, _, mem = wrap(xm, xs, mem)
torch.cuda.synchronize()
print(f"GRU 144/144, 384 columns: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per step")
