"""Offline CNN baseline forward (row a16): implicit-GEMM HIP path vs the torch restatement of the Keras model.
Parity UNPINNED by reference artefacts (model/saved_model.pb ships without variables; TF absent): random weights."""
import numpy as np
import pytest
import torch

from oracle import torch_ref


def _arch(depth, width, seed=0):
    g = torch.Generator().manual_seed(seed)
    ws, bs = [], []

    def conv(co, ci, k):
        ws.append(torch.randn(co, ci, k, generator=g) * (0.9 / np.sqrt(ci * k)))
        bs.append(torch.randn(co, generator=g) * 0.05)
    for i in range(depth):
        ci = 6 if i == 0 else width
        conv(width, ci, 3); conv(width, width, 3); conv(width, ci, 1)
    conv(10, width, 1)
    conv(10, 10, 1)
    return ws, bs


def test_flop_count_matches_survey():
    # ~26.4 MFLOP per level, 1.58 GFLOP per column
    per_level = 2 * (3 * 6 * 406 + 3 * 406 * 406 + 6 * 406) + 11 * 2 * (2 * 3 * 406 * 406 + 406 * 406) + 2 * 406 * 10 + 2 * 100
    assert abs(per_level * 60 / 1.58e9 - 1) < 0.03


@pytest.mark.gpu
@pytest.mark.parametrize("depth,width,B", [(2, 40, 3), (12, 406, 5), (12, 406, 37)])
def test_cnn_forward_matches_torch(depth, width, B):
    from climsim_amd.baselines import CNNBaseline
    ws, bs = _arch(depth, width)
    m = CNNBaseline([w.numpy() for w in ws], [b.numpy() for b in bs], depth=depth, width=width, max_batch=64)
    x = torch.randn(B, 60, 6, generator=torch.Generator().manual_seed(B))
    y = m(x.cuda()).cpu()
    ref = torch_ref.cnn_ref(x.double(), [w.double() for w in ws], [b.double() for b in bs], depth=depth)
    ref32 = torch_ref.cnn_ref(x, ws, bs, depth=depth)
    scale = ref.abs().max().item()
    e = (y.double() - ref).abs().max().item()
    assert e <= 1e-5 * scale, e / scale
    assert e <= 4 * (ref32.double() - ref).abs().max().item() + 1e-7 * scale
    assert (y[:, :, 2:] >= 0).all()
