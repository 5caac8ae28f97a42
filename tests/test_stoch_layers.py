"""Stochastic recurrent layers (row a9) against golden vectors produced by the reference's own layer classes
(tests/golden/make_golden_stoch.py), with the reference's N(0,1) draw stored and replayed."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from oracle import torch_ref

G = np.load(os.path.join(GOLDEN, "stoch_layers.npz"))
t = lambda k: torch.from_numpy(G[k])


@pytest.mark.parametrize("B", [8, 5])
def test_torch_restatements_bitexact_vs_reference_layers(B):
    for tag in ("gru5", "gru5b"):
        out = torch_ref.stoch_gru5_ref(t(f"B{B}.x"), t(f"B{B}.h0"), t(f"B{B}.{tag}.eps"), t(f"{tag}.w.weight_ih"),
                                       t(f"{tag}.w.weight_zh"), t(f"{tag}.w.weight_encoder"),
                                       t(f"{tag}.w.bias_ih") if f"{tag}.w.bias_ih" in G.files else None,
                                       t(f"{tag}.w.bias_zh") if f"{tag}.w.bias_zh" in G.files else None)
        assert rel_err(out.numpy(), G[f"B{B}.{tag}.out"]) <= 1e-6
    out, (hT, cT) = torch_ref.stoch_lstm4_ref(t(f"B{B}.x"), t(f"B{B}.h0"), t(f"B{B}.c0"), t(f"B{B}.lstm4.eps"),
                                              t("lstm4.w.weight_encoder"))
    assert rel_err(out.numpy(), G[f"B{B}.lstm4.out"]) <= 1e-6
    assert rel_err(cT.numpy(), G[f"B{B}.lstm4.cT"]) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B", [8, 5])
def test_hip_stochastic_gru5_vs_reference(B):
    from climsim_amd.layers import MyStochasticGRULayer5
    for tag in ("gru5", "gru5b"):
        bi = G[f"{tag}.w.bias_ih"] if f"{tag}.w.bias_ih" in G.files else None
        bz = G[f"{tag}.w.bias_zh"] if f"{tag}.w.bias_zh" in G.files else None
        layer = MyStochasticGRULayer5(G[f"{tag}.w.weight_ih"], G[f"{tag}.w.weight_zh"], G[f"{tag}.w.weight_encoder"], bi, bz,
                                      max_rows=60 * 16)
        out = layer(t(f"B{B}.x").cuda(), t(f"B{B}.h0").cuda(), eps=t(f"B{B}.{tag}.eps").cuda())
        assert rel_err(out.cpu().numpy(), G[f"B{B}.{tag}.out"]) <= 1e-5, tag


@pytest.mark.gpu
@pytest.mark.parametrize("B", [8, 5])
def test_hip_stochastic_lstm4_vs_reference(B):
    from climsim_amd.layers import MyStochasticLSTMLayer4
    layer = MyStochasticLSTMLayer4(G["lstm4.w.weight_encoder"], 128, max_rows=60 * 16)
    out, (hT, cT) = layer(t(f"B{B}.x").cuda(), (t(f"B{B}.h0").cuda(), t(f"B{B}.c0").cuda()), eps=t(f"B{B}.lstm4.eps").cuda())
    assert rel_err(out.cpu().numpy(), G[f"B{B}.lstm4.out"]) <= 1e-5
    assert rel_err(hT.cpu().numpy(), G[f"B{B}.lstm4.hT"]) <= 1e-5
    assert rel_err(cT.cpu().numpy(), G[f"B{B}.lstm4.cT"]) <= 1e-5


GR = np.load(os.path.join(GOLDEN, "stoch_grads.npz"))
tg = lambda k: torch.from_numpy(GR[k])


@pytest.mark.gpu
@pytest.mark.parametrize("B", [6, 3])
@pytest.mark.parametrize("tag", ["gru5", "gru5b"])
def test_hip_stochastic_gru5_backward_vs_reference_autograd(tag, B):
    """The HIP BPTT of MyStochasticGRULayer5 (the reference's hand-written CUDA backward, :85-126, :176-232) against autograd
    through the reference class (make_golden_stoch.py), through the same plugin interface: a torch.autograd.Function."""
    from climsim_amd.layers import MyStochasticGRULayer5
    w = {k: G[f"{tag}.w.{k}"] for k in ("weight_ih", "weight_zh", "weight_encoder")}
    bi = G[f"{tag}.w.bias_ih"] if f"{tag}.w.bias_ih" in G.files else None
    bz = G[f"{tag}.w.bias_zh"] if f"{tag}.w.bias_zh" in G.files else None
    layer = MyStochasticGRULayer5(w["weight_ih"], w["weight_zh"], w["weight_encoder"], bi, bz, max_rows=60 * 8, requires_grad=True)
    x = tg(f"B{B}.x").cuda().requires_grad_(True)
    h0 = tg(f"B{B}.h0").cuda().requires_grad_(True)
    out = layer(x, h0, eps=tg(f"B{B}.{tag}.eps").cuda())
    assert rel_err(out.detach().cpu().numpy(), GR[f"B{B}.{tag}.out"]) <= 1e-5
    (out * tg(f"B{B}.dout").cuda()).sum().backward()
    assert rel_err(x.grad.cpu().numpy(), GR[f"B{B}.{tag}.dx"]) <= 2e-5
    assert rel_err(h0.grad.cpu().numpy(), GR[f"B{B}.{tag}.dh0"]) <= 2e-5
    for n, p in layer.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), GR[f"B{B}.{tag}.dw.{n}"]) <= 2e-5, n


@pytest.mark.gpu
@pytest.mark.parametrize("B", [6, 3])
def test_hip_stochastic_lstm4_backward_vs_reference_autograd(B):
    from climsim_amd.layers import MyStochasticLSTMLayer4
    layer = MyStochasticLSTMLayer4(G["lstm4.w.weight_encoder"], 128, max_rows=60 * 8, requires_grad=True)
    x = tg(f"B{B}.x").cuda().requires_grad_(True)
    h0 = tg(f"B{B}.h0").cuda().requires_grad_(True)
    c0 = tg(f"B{B}.c0").cuda().requires_grad_(True)
    out, (hT, cT) = layer(x, (h0, c0), eps=tg(f"B{B}.lstm4.eps").cuda())
    assert rel_err(out.detach().cpu().numpy(), GR[f"B{B}.lstm4.out"]) <= 1e-5
    ((out * tg(f"B{B}.dout").cuda()).sum() + (hT * tg(f"B{B}.dhT").cuda()).sum() + (cT * tg(f"B{B}.dcT").cuda()).sum()).backward()
    assert rel_err(x.grad.cpu().numpy(), GR[f"B{B}.lstm4.dx"]) <= 2e-5
    assert rel_err(h0.grad.cpu().numpy(), GR[f"B{B}.lstm4.dh0"]) <= 2e-5
    assert rel_err(c0.grad.cpu().numpy(), GR[f"B{B}.lstm4.dc0"]) <= 2e-5
    assert rel_err(layer.weight_encoder.grad.cpu().numpy(), GR[f"B{B}.lstm4.dw.weight_encoder"]) <= 2e-5
    # gradients accumulate over calls and a second identical pass is bit-identical (no atomics)
    g1 = layer.weight_encoder.grad.clone()
    layer.weight_encoder.grad = None
    x.grad = None
    out2, (hT2, cT2) = layer(x, (h0, c0), eps=tg(f"B{B}.lstm4.eps").cuda())
    ((out2 * tg(f"B{B}.dout").cuda()).sum() + (hT2 * tg(f"B{B}.dhT").cuda()).sum() + (cT2 * tg(f"B{B}.dcT").cuda()).sum()).backward()
    assert torch.equal(layer.weight_encoder.grad, g1)
    # inference mode is unchanged by the training plumbing
    with torch.no_grad():
        out3, _ = layer(x, (h0, c0), eps=tg(f"B{B}.lstm4.eps").cuda())
    assert torch.equal(out3, out.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["gru5b", "lstm4"])
def test_pending_forwards_keep_their_own_activations(kind):
    """The reference's TBPTT use of the autograd layer: several forwards of the SAME layer (window steps, a no_grad validation
    forward in between) and THEN the backwards.  Every forward owns its saved activations (advisor finding, round 2): the gradients
    of the first forward, taken after a second forward on other inputs and an inference call, equal the reference's autograd
    goldens of that first forward; a second backward through a consumed forward raises; a backward after the weights were re-packed
    raises; an in-place parameter update re-packs the kernels' weight copies by itself."""
    from climsim_amd.layers import MyStochasticGRULayer5, MyStochasticLSTMLayer4
    B, B2 = 6, 3
    dev = lambda k: tg(k).cuda()
    if kind == "lstm4":
        layer = MyStochasticLSTMLayer4(G["lstm4.w.weight_encoder"], 128, max_rows=60 * 8, requires_grad=True)

        def run(Bv, grad=True):
            x, h0, c0 = (dev(f"B{Bv}.{n}").requires_grad_(grad) for n in ("x", "h0", "c0"))
            out, (hT, cT) = layer(x, (h0, c0), eps=dev(f"B{Bv}.lstm4.eps"))
            return (x, h0, c0), out, (out * dev(f"B{Bv}.dout")).sum() + (hT * dev(f"B{Bv}.dhT")).sum() + (cT * dev(f"B{Bv}.dcT")).sum()
        names = ("dx", "dh0", "dc0")
    else:
        w = {k: G[f"{kind}.w.{k}"] for k in ("weight_ih", "weight_zh", "weight_encoder", "bias_ih", "bias_zh")}
        layer = MyStochasticGRULayer5(w["weight_ih"], w["weight_zh"], w["weight_encoder"], w["bias_ih"], w["bias_zh"], max_rows=60 * 8, requires_grad=True)

        def run(Bv, grad=True):
            x, h0 = (dev(f"B{Bv}.{n}").requires_grad_(grad) for n in ("x", "h0"))
            out = layer(x, h0, eps=dev(f"B{Bv}.{kind}.eps"))
            return (x, h0), out, (out * dev(f"B{Bv}.dout")).sum()
        names = ("dx", "dh0")
    leaves1, out1, loss1 = run(B)
    leaves2, out2, loss2 = run(B2)                    # a second forward of the same layer, other batch
    with torch.no_grad():
        run(B, grad=False)                            # and an inference forward: neither may disturb the first one's activations
    loss1.backward(retain_graph=True)
    for leaf, n in zip(leaves1, names):
        assert rel_err(leaf.grad.cpu().numpy(), GR[f"B{B}.{kind}.{n}"]) <= 2e-5, n
    for n, p in layer.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), GR[f"B{B}.{kind}.dw.{n}"]) <= 2e-5, n
        p.grad = None
    loss2.backward()                                  # the second forward's backward still has ITS activations
    for leaf, n in zip(leaves2, names):
        assert rel_err(leaf.grad.cpu().numpy(), GR[f"B{B2}.{kind}.{n}"]) <= 2e-5, n
    for n, p in layer.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), GR[f"B{B2}.{kind}.dw.{n}"]) <= 2e-5, n
    with pytest.raises(RuntimeError, match="consumed"):
        loss1.backward()
    # an in-place parameter update (what optimizer.step() does) is picked up by the next forward without sync_params()
    _, out_a, loss_a = run(B)
    with torch.no_grad():
        for p in layer.parameters():
            p.mul_(0.5)
    _, out_b, _ = run(B)
    assert not torch.equal(out_a, out_b)
    # ... and what it then computes is what a layer built from the updated parameters computes
    if kind == "lstm4":
        fresh = MyStochasticLSTMLayer4(layer.weight_encoder.detach(), 128, max_rows=60 * 8)
        out_c, _ = fresh(dev(f"B{B}.x"), (dev(f"B{B}.h0"), dev(f"B{B}.c0")), eps=dev(f"B{B}.lstm4.eps"))
    else:
        fresh = MyStochasticGRULayer5(*[p.detach() for p in layer._param_list()], max_rows=60 * 8)
        out_c = fresh(dev(f"B{B}.x"), dev(f"B{B}.h0"), eps=dev(f"B{B}.{kind}.eps"))
    assert torch.equal(out_b.detach(), out_c)
    with pytest.raises(RuntimeError, match="re-packed"):
        loss_a.backward()
