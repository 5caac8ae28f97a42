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
