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


@pytest.mark.gpu
@pytest.mark.parametrize("depth,width,B,dropout", [(2, 40, 3, 0.0), (3, 72, 4, 0.175), (3, 406, 6, 0.175), (6, 406, 6, 0.0),
                                                  (12, 200, 6, 0.0), (12, 406, 6, 0.175)])
def test_cnn_training_step_matches_autograd(depth, width, B, dropout):
    """forward (with the caller's dropout masks), mae_adjusted, every weight / bias gradient and one Keras-Adam update
    against torch autograd on the restatement (fp32, the arithmetic the reference trains in).

    The gradient of a ReLU network is piecewise constant in the activations: a unit whose pre-activation is within
    fp32 rounding of zero can take a different branch in two correct fp32 evaluations (torch fp32 on two CPUs, or vs
    torch fp64, differ the same way: tests/reports/cnn_dbg.py), and with B*60 = 360 rows ONE flipped unit moves a
    weight-gradient row by ~1/sqrt(360) of its size.  The forward is therefore checked against the plain restatement,
    and the gradients against the restatement evaluated with the ReLU branch pattern of the HIP forward (read back
    through the csa_cnn_train_get_act tap); the patterns themselves must agree except at a handful of units."""
    from climsim_amd.baselines import CNNTrainer
    ws, bs = _arch(depth, width, seed=3)
    tr = CNNTrainer([w.numpy() for w in ws], [b.numpy() for b in bs], depth=depth, width=width, dropout=dropout, max_batch=8)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 60, 6, generator=g)
    yt = torch.randn(B, 60, 10, generator=g)
    masks = None
    if dropout > 0:
        masks = (torch.rand(2 * depth, B * 60, width, generator=g) >= dropout).to(torch.uint8)
    y = tr.forward(x.cuda(), None if masks is None else masks.cuda()).cpu()
    loss, grads = tr.backward(yt.cuda())
    # reference: fp64 autograd
    wd = [w.clone().requires_grad_(True) for w in ws]
    bd = [b.clone().requires_grad_(True) for b in bs]
    mk = None if masks is None else [m.view(B, 60, width).float() for m in masks]
    with torch.no_grad():
        y_plain = torch_ref.cnn_ref(x, ws, bs, depth=depth, masks=mk, dropout=dropout)
    assert (y - y_plain).abs().max().item() <= 1e-5 * y_plain.abs().max().item()
    gates = []
    for blk in range(depth):
        for which in (0, 1):
            t = tr.saved_activation(blk, which, B).cpu()
            gates.append((t > 0).float() if masks is None else ((t > 0) | (masks[2 * blk + which].view(B, 60, width) == 0)).float())
    yr = torch_ref.cnn_ref(x, wd, bd, depth=depth, masks=mk, dropout=dropout, gates=gates)
    # branch patterns agree except at units within rounding of zero: outputs identical to rounding
    assert (yr.detach() - y_plain).abs().max().item() <= 1e-5 * y_plain.abs().max().item()
    lr_ = torch_ref.mae_adjusted(yt, yr)
    lr_.backward()
    scale = yr.abs().max().item()
    assert (y.double() - yr.detach()).abs().max().item() <= 1e-5 * scale
    assert abs(loss.item() - lr_.item()) <= 1e-5 * abs(lr_.item())
    gw, gb = tr.unpack(grads)
    for name, got, refs in (("w", gw, wd), ("b", gb, bd)):
        for i, (a, r) in enumerate(zip(got, refs)):
            ref = r.grad.double()
            err = (a.double() - ref).abs()
            assert err.max().item() <= 2e-5 * ref.abs().max().item() + 1e-12, (name, i)
    # padding entries of the flat gradient are exactly zero
    total = sum(a.numel() for a in gw) + sum(a.numel() for a in gb)
    assert int((grads != 0).sum().item()) <= total
    # one Keras-Adam step: p -= lr*sqrt(1-b2)/(1-b1) * m/(sqrt(v)+eps) with m=(1-b1)g, v=(1-b2)g^2
    p0w, p0b = tr.unpack(tr.flat_params())
    tr.adam(lr=1e-3)
    p1w, p1b = tr.unpack(tr.flat_params())
    for p0, p1, gq in zip(p0w + p0b, p1w + p1b, gw + gb):
        gq = gq.double()
        m, v = 0.1 * gq, 0.001 * gq * gq
        upd = 1e-3 * (1 - 0.999) ** 0.5 / (1 - 0.9) * m / (v.sqrt() + 1e-7)
        assert ((p0.double() - p1.double()) - upd).abs().max().item() <= 2e-6 * 1e-3 + 1e-4 * upd.abs().max().item()


def _np_reshape_to(a, nscal):      # climsim_utils/data_utils.py:2104-2150 restated (test infrastructure)
    return np.stack([a[:, 0:60], a[:, 60:120]] + [np.repeat(a[:, 120 + i][:, None], 60, axis=1) for i in range(nscal)], axis=2)


def _np_reshape_from(p):           # :2152-2175
    return np.concatenate([p[:, :, 0], p[:, :, 1]] + [np.mean(p[:, :, c], axis=1)[:, None] for c in range(2, 10)], axis=1)


@pytest.mark.gpu
def test_cnn_data_format_adapters():
    from climsim_amd.data_utils import data_utils
    g = np.random.Generator(np.random.PCG64(4))
    x = g.standard_normal((37, 124)).astype(np.float32)
    t = g.standard_normal((37, 128)).astype(np.float32)
    p = g.standard_normal((37, 60, 10)).astype(np.float32)
    d = lambda a: torch.from_numpy(a).cuda()
    assert np.array_equal(data_utils.reshape_input_for_cnn(d(x)).cpu().numpy(), _np_reshape_to(x, 4))
    assert np.array_equal(data_utils.reshape_target_for_cnn(d(t)).cpu().numpy(), _np_reshape_to(t, 8))
    back = data_utils.reshape_target_from_cnn(d(p)).cpu().numpy()
    ref = _np_reshape_from(p)
    assert np.array_equal(back[:, :120], ref[:, :120])
    assert np.abs(back[:, 120:] - ref[:, 120:]).max() <= 2e-7        # level mean: summation order only
    # round trip: scalars repeated over the levels average back exactly
    assert np.abs(data_utils.reshape_target_from_cnn(data_utils.reshape_target_for_cnn(d(t))).cpu().numpy() - t).max() <= 1e-6 * np.abs(t).max()


@pytest.mark.gpu
def test_cnn_training_step_at_the_config3_shard_size():
    """BASELINE.json configs[3]: the 21,600-column high-res grid sharded 8-way = 2,700 columns per GPU, full architecture
    (12 blocks x 406 channels, Dropout 0.175).  Size-independent properties at full size: bitwise determinism, finite loss,
    padding gradients exactly zero, and the data-parallel property (two half shards with grad_scale 1/2 each sum to the
    unsharded flat gradient: one all-reduce is all a step needs)."""
    from climsim_amd.baselines import CNNTrainer
    B, depth, width = 2700, 12, 406
    ws, bs = _arch(depth, width, seed=5)
    tr = CNNTrainer([w.numpy() for w in ws], [b.numpy() for b in bs], depth=depth, width=width, dropout=0.175, max_batch=B)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, 60, 6, generator=g).cuda()
    yt = torch.randn(B, 60, 10, generator=g).cuda()
    masks = tr.draw_masks(B)
    y1 = tr.forward(x, masks).clone()
    loss1, g1 = tr.backward(yt)
    loss1, g1 = loss1.clone(), g1.clone()
    y2 = tr.forward(x, masks)
    loss2, g2 = tr.backward(yt)
    assert torch.equal(y1, y2) and torch.equal(g1, g2) and torch.equal(loss1, loss2)          # no atomics anywhere
    assert torch.isfinite(y1).all() and torch.isfinite(g1).all() and float(loss1) > 0
    gw, gb = tr.unpack(g1)
    assert int((g1 != 0).sum().item()) <= sum(a.numel() for a in gw) + sum(a.numel() for a in gb)
    # data parallel: halves of the batch, each with grad_scale = its share of the global mean
    h = B // 2
    m3 = masks.view(2 * depth, B, 60, -1)
    gsum, lsum = torch.zeros_like(g1), 0.0
    for lo, hi in ((0, h), (h, B)):
        tr.forward(x[lo:hi].contiguous(), m3[:, lo:hi].reshape(2 * depth, (hi - lo) * 60, -1).contiguous())
        l, gpart = tr.backward(yt[lo:hi].contiguous(), (hi - lo) / B)
        gsum += gpart
        lsum += float(l)
    assert float((gsum - g1).abs().max()) <= 2e-6 * float(g1.abs().max())
    assert abs(lsum - float(loss1)) <= 1e-5 * float(loss1)
    # one optimiser step changes the parameters and keeps them finite
    p0 = tr.flat_params()
    tr.adam(lr=1e-4)
    p1 = tr.flat_params()
    assert torch.isfinite(p1).all() and not torch.equal(p0, p1)
