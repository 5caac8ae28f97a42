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


@pytest.mark.gpu
@pytest.mark.parametrize("B", [37, 384])
def test_mlp_training_step_matches_autograd(B):
    """forward, 'mse' loss, every weight / bias gradient and one Keras-Adam update against torch autograd on the restatement
    (LeakyReLU has a kink at zero: a unit within fp32 rounding of it can take the other slope in two correct evaluations,
    so 0.1 % of a tensor's entries may exceed the tolerance as long as its relative L2 error stays below 1e-3)."""
    from climsim_amd.baselines import MLPTrainer
    dims, ws, bs = _arch()
    tr = MLPTrainer([w.numpy() for w in ws], [b.numpy() for b in bs], max_batch=512)
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 124, generator=g)
    yt = torch.randn(B, 128, generator=g)
    y = tr.forward(x.cuda()).cpu()
    loss, grads = tr.backward(yt.cuda())
    wd = [w.clone().requires_grad_(True) for w in ws]
    bd = [b.clone().requires_grad_(True) for b in bs]
    yr = torch_ref.mlp_ref(x, wd, bd)
    lr_ = torch.mean((yr - yt) ** 2)
    lr_.backward()
    assert (y - yr.detach()).abs().max().item() <= 1e-5 * yr.abs().max().item()
    assert abs(loss.item() - lr_.item()) <= 1e-5 * lr_.item()
    gw, gb = tr.unpack(grads)
    for name, got, refs in (("w", gw, wd), ("b", gb, bd)):
        for i, (a, r) in enumerate(zip(got, refs)):
            ref = r.grad
            err = (a - ref).abs()
            tol = 2e-5 * ref.abs().max().item() + 1e-12
            assert (err > tol).float().mean().item() <= 1e-3, (name, i, err.max().item() / ref.abs().max().item())
            assert err.norm().item() <= 1e-3 * ref.norm().item() + 1e-12, (name, i)
    p0 = tr.flat_params()
    tr.adam(lr=1e-3)
    p1 = tr.flat_params()
    gq = grads.double()
    upd = 1e-3 * (1 - 0.999) ** 0.5 / (1 - 0.9) * (0.1 * gq) / ((0.001 * gq * gq).sqrt() + 1e-7)
    assert ((p0.double() - p1.double()) - upd).abs().max().item() <= 2e-6 * 1e-3 + 1e-4 * upd.abs().max().item()
