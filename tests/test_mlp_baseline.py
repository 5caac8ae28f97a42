"""Offline MLP baseline (row a15): HIP GEMM chain vs the torch fp32 restatement of the Keras model.
Parity is UNPINNED by reference artefacts (no weights / outputs ship with the reference; TF is absent):
random weights of the published architecture, 1,753,472 parameters."""
import numpy as np
import pytest
import torch

from oracle import torch_ref


def _arch(seed=0):
    g = torch.Generator().manual_seed(seed)
    dims = [124, 768, 640, 512, 640, 640, 128, 128]
    ws, bs = [], []
    for a, b in zip(dims[:-1], dims[1:]):
        ws.append(torch.randn(b, a, generator=g) * (1.0 / np.sqrt(a)))
        bs.append(torch.randn(b, generator=g) * 0.1)
    return dims, ws, bs


def test_parameter_count_matches_reference_notebook():
    dims, ws, bs = _arch()
    assert sum(w.numel() for w in ws) + sum(b.numel() for b in bs) == 1753472   # FLOP_calculation.ipynb cell 5


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 37, 384, 2700])
def test_mlp_forward_matches_torch(B):
    from climsim_amd.baselines import MLPBaseline
    dims, ws, bs = _arch()
    m = MLPBaseline([w.numpy() for w in ws], [b.numpy() for b in bs], max_batch=3000)
    x = torch.randn(B, 124, generator=torch.Generator().manual_seed(B))
    y = m(x.cuda()).cpu()
    ref = torch_ref.mlp_ref(x.double(), [w.double() for w in ws], [b.double() for b in bs])
    ref32 = torch_ref.mlp_ref(x, ws, bs)
    scale = ref.abs().max().item()
    assert (y.double() - ref).abs().max().item() <= 1e-5 * scale
    # and no worse than torch's own fp32 evaluation
    assert (y.double() - ref).abs().max().item() <= 4 * (ref32.double() - ref).abs().max().item() + 1e-7 * scale
    assert (y[:, 120:] >= 0).all()
