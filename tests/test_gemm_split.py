"""The opt-in split-bf16 projection GEMM (gemm.hip::proj_gemm_b3_kernel, csa_set_gemm_split) against a float64 product, next to the
default fp32 MFMA chain on the same inputs: reference rnn/models/models.py:493,536 (nn.LSTM's W_ih x + b for all levels at once).
The split writes every fp32 operand exactly as three bf16 values and accumulates six partial products in fp32; the dropped terms are
<= 2^-24 |a||b| each.  The test holds it to the fp32 chain's own error against float64 (tolerance stated below), at the benchmarked
shape (K = 128, full tiles), with K = 144 (nine 16-deep chunks: the odd-chunk tail) and with a ragged row count (partial tiles)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _model(tag):
    d = np.load(os.path.join(ROOT, "tests", "golden", f"{tag}_model.npz"))
    return ({k[2:]: d[k] for k in d.files if k.startswith("c.")}, {k[2:]: d[k] for k in d.files if k.startswith("w.")})


@pytest.mark.parametrize("tag,stage,B", [("v4_stateless", 1, 384), ("v4_stateless", 3, 384), ("v4_memory", 1, 384), ("v4_memory", 1, 193)])
def test_split_gemm_is_as_close_to_float64_as_the_fp32_chain(tag, stage, B):
    import torch
    import climsim_amd
    from climsim_amd import _lib
    consts, weights = _model(tag)
    model = climsim_amd.NewModel_constraint(consts, weights, max_batch=B)
    em, cfg = model.emulator, model.emulator.cfg
    L, nh, nm = cfg.nlev, cfg.nh1, cfg.nh_mem
    K = nh + nm if stage == 1 else nh
    rnn = "rnn1" if stage == 1 else "rnn2"
    rng = np.random.Generator(np.random.PCG64(11 * stage + B))
    X = (rng.standard_normal((L * B, K)) * np.exp(rng.standard_normal((L * B, 1)))).astype(np.float32)
    W = weights[f"{rnn}.weight_ih_l0"].astype(np.float64)
    b = weights[f"{rnn}.bias_ih_l0"].astype(np.float64) + weights[f"{rnn}.bias_hh_l0"].astype(np.float64)
    assert W.shape == (4 * nh, K)
    R = X.astype(np.float64) @ W.T + b
    mag = np.abs(X.astype(np.float64)) @ np.abs(W.T) + np.abs(b)           # sum |a||b| per output: the scale of a rounding error
    dX = torch.from_numpy(X).cuda()
    out = {}
    try:
        for mode in (0, 1):
            _lib.lib().csa_set_gemm_split(mode)
            (P,) = em.debug_stage(stage, B, [dX], [(L * B, 4 * nh)])
            torch.cuda.synchronize()
            out[mode] = P.cpu().numpy().astype(np.float64)
    finally:
        _lib.lib().csa_set_gemm_split(0)
        em.close()
    # the library's unit-major gate order -> the reference's column order, found on the fp32 chain's output
    perm = np.array([np.argmin(np.abs(R[:64, :] - out[0][:64, c:c + 1]).sum(0)) for c in range(4 * nh)])
    assert len(set(perm.tolist())) == 4 * nh
    Rp, magp = R[:, perm], mag[:, perm]
    err = {m: np.abs(out[m] - Rp) for m in (0, 1)}
    rel = {m: float(np.max(err[m] / magp)) for m in (0, 1)}
    rms = {m: float(np.sqrt((err[m] ** 2).mean())) for m in (0, 1)}
    print(f"{tag} stage {stage} B {B} K {K}: max err / sum|a||b|  chain {rel[0]:.2e}  split {rel[1]:.2e};  rms err  chain {rms[0]:.3e}  split {rms[1]:.3e}")
    # tolerance: a few fp32 roundings relative to sum|a||b| (2^-24 = 6e-8; the chain makes K of them in sequence), and never worse than
    # 1.25 x the fp32 chain's own rms error against float64
    assert rel[1] < 1e-6
    assert rms[1] <= 1.25 * rms[0]
