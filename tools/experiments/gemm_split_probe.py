"""Experiment (record): the split-bf16 projection GEMM (gemm.hip::proj_gemm_b3_kernel, csa_set_gemm_split) against the fp32 MFMA
chain, both against a float64 product: error of each, and HIP-event time per launch, at the headline shape (23,040 x 144 -> 512).
Run on the GPU box: python tools/experiments/gemm_split_probe.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, climsim_amd
from climsim_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
consts, weights = bench.load_model(bench.WORKLOADS["v4_stateless_384"][0])
model = climsim_amd.NewModel_constraint(consts, weights, max_batch=B)
em, cfg = model.emulator, model.emulator.cfg
L, nh, nm = cfg.nlev, cfg.nh1, cfg.nh_mem
rng = np.random.Generator(np.random.PCG64(5))
for stage, K, wk, bk in ((1, nh + nm, "rnn1.weight_ih_l0", ("rnn1.bias_ih_l0", "rnn1.bias_hh_l0")), (3, nh, "rnn2.weight_ih_l0", ("rnn2.bias_ih_l0", "rnn2.bias_hh_l0"))):
    X = (rng.standard_normal((L * B, K)) * np.exp(rng.standard_normal((L * B, 1)))).astype(np.float32)
    dX = torch.from_numpy(X).cuda()
    W = weights[wk].astype(np.float64); b = sum(weights[k].astype(np.float64) for k in bk)
    R = X.astype(np.float64) @ W.T + b                       # natural column order
    outs = {}
    for mode in (0, 1):
        _lib.lib().csa_set_gemm_split(mode)
        (P,) = em.debug_stage(stage, B, [dX], [(L * B, 4 * nh)])
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(20): em.debug_stage(stage, B, [dX], [(L * B, 4 * nh)])
        ev[0].record()
        for _ in range(200): em.debug_stage(stage, B, [dX], [(L * B, 4 * nh)])
        ev[1].record(); torch.cuda.synchronize()
        outs[mode] = (P.cpu().numpy().astype(np.float64), ev[0].elapsed_time(ev[1]) / 200 * 1e3)
    # packed column -> natural column, found numerically on the fp32 chain's output
    P0 = outs[0][0]
    perm = np.array([np.argmin(np.abs(R[:64, :] - P0[:64, c:c + 1]).sum(0)) for c in range(4 * nh)])
    Rp = R[:, perm]
    scale = np.abs(Rp).max()
    mag = np.abs(X.astype(np.float64)) @ np.abs(W.T)[:, perm] + np.abs(b)[perm]      # sum |a||b|: the natural error scale
    for mode in (0, 1):
        P, us = outs[mode]
        e = np.abs(P - Rp)
        print(f"stage {stage} K {K} split {mode}: {us:7.2f} us per launch (incl. launch overhead of the debug entry); max|err| {e.max():.3e} = {e.max() / scale:.2e} of max|ref|; "
              f"max err / sum|a||b| {np.max(e / mag):.3e}; rms err / rms ref {np.sqrt((e ** 2).mean()) / np.sqrt((Rp ** 2).mean()):.3e}")
    print(f"  split vs chain: max|diff| {np.abs(outs[1][0] - outs[0][0]).max():.3e}")
_lib.lib().csa_set_gemm_split(0)
