#!/usr/bin/env python3
"""Golden vectors of the reference's stochastic recurrent layers:
rnn/models_torch_kernels.py::MyStochasticGRULayer5 (:834-891, the CPU branch of the layer whose CUDA path
is the repository's only native code, :29-252) and ::MyStochasticLSTMLayer4 (:1474-1531).
Both draw eps = randn(T,B,H) at the top of forward; the draw is reproduced under the same seed and stored,
so the HIP kernels can be fed the identical noise.  Run with TORCHDYNAMO_DISABLE=1 (LSTMLayer4.forward is
decorated with torch.compile; eager execution keeps the eager RNG stream).  Build container only."""
import os, sys, types
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
import numpy as np
import torch

REF = "/root/reference/rnn"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    stub = types.ModuleType("omegaconf"); stub.DictConfig = dict; stub.OmegaConf = object
    sys.modules["omegaconf"] = stub
    import torch.utils.cpp_extension as ce
    ce.load_inline = lambda *a, **k: None
    sys.path.insert(0, REF)
    import models_torch_kernels as K
    d = {}
    T, nx, H = 60, 128, 128
    torch.manual_seed(2024)
    gru = K.MyStochasticGRULayer5(nx, H, use_bias=False)
    gru_b = K.MyStochasticGRULayer5(nx, H, use_bias=True)
    lstm = K.MyStochasticLSTMLayer4(nx, H, use_bias=False)
    for tag, m in (("gru5", gru), ("gru5b", gru_b), ("lstm4", lstm)):
        for n, p in m.named_parameters():
            d[f"{tag}.w.{n}"] = p.detach().numpy().astype(np.float32)
    for B, seed in ((8, 1), (5, 2)):
        x = torch.randn(T, B, nx) * 0.7
        h0 = torch.randn(B, H) * 0.5
        c0 = torch.randn(B, H) * 0.5
        d[f"B{B}.x"], d[f"B{B}.h0"], d[f"B{B}.c0"] = x.numpy(), h0.numpy(), c0.numpy()
        for tag, m in (("gru5", gru), ("gru5b", gru_b)):
            torch.manual_seed(100 + seed)
            with torch.no_grad():
                out = m(x, h0)
            torch.manual_seed(100 + seed)
            d[f"B{B}.{tag}.eps"] = torch.randn(T, B, H).numpy()
            d[f"B{B}.{tag}.out"] = out.numpy()
        torch.manual_seed(200 + seed)
        with torch.no_grad():
            out, (hT, cT) = lstm(x, (h0, c0))
        torch.manual_seed(200 + seed)
        d[f"B{B}.lstm4.eps"] = torch.randn(T, B, H).numpy()
        d[f"B{B}.lstm4.out"], d[f"B{B}.lstm4.hT"], d[f"B{B}.lstm4.cT"] = out.numpy(), hT.numpy(), cT.numpy()
        print(B, float(out.abs().max()))
    np.savez_compressed(f"{OUT}/stoch_layers.npz", **d)

    # ---- gradients: autograd through the reference layer classes (the GRU's CPU branch is plain torch; its CUDA branch is the
    # hand-written backward of :85-126,:176-232 that the HIP BPTT kernels replace).  Loss = <out, dout> (+ <hT, dhT> + <cT, dcT>).
    g = {}
    for B, seed in ((6, 11), (3, 12)):
        gen = torch.Generator().manual_seed(seed)
        x = (torch.randn(T, B, nx, generator=gen) * 0.7).requires_grad_(True)
        h0 = (torch.randn(B, H, generator=gen) * 0.5).requires_grad_(True)
        c0 = (torch.randn(B, H, generator=gen) * 0.5).requires_grad_(True)
        dout = torch.randn(T, B, H, generator=gen)
        dhT, dcT = torch.randn(B, H, generator=gen), torch.randn(B, H, generator=gen)
        g[f"B{B}.x"], g[f"B{B}.h0"], g[f"B{B}.c0"] = x.detach().numpy(), h0.detach().numpy(), c0.detach().numpy()
        g[f"B{B}.dout"], g[f"B{B}.dhT"], g[f"B{B}.dcT"] = dout.numpy(), dhT.numpy(), dcT.numpy()
        for tag, m in (("gru5", gru), ("gru5b", gru_b), ("lstm4", lstm)):
            for t_ in (x, h0, c0):
                t_.grad = None
            m.zero_grad()
            torch.manual_seed(300 + seed)
            if tag == "lstm4":
                out, (hT, cT) = m(x, (h0, c0))
                loss = (out * dout).sum() + (hT * dhT).sum() + (cT * dcT).sum()
            else:
                out = m(x, h0)
                loss = (out * dout).sum()
            loss.backward()
            torch.manual_seed(300 + seed)
            g[f"B{B}.{tag}.eps"] = torch.randn(T, B, H).numpy()
            g[f"B{B}.{tag}.out"] = out.detach().numpy()
            g[f"B{B}.{tag}.dx"], g[f"B{B}.{tag}.dh0"] = x.grad.numpy().copy(), h0.grad.numpy().copy()
            if tag == "lstm4":
                g[f"B{B}.{tag}.dc0"] = c0.grad.numpy().copy()
            for n, p_ in m.named_parameters():
                g[f"B{B}.{tag}.dw.{n}"] = p_.grad.numpy().copy()
            print("grad", B, tag, float(x.grad.abs().max()))
    np.savez_compressed(f"{OUT}/stoch_grads.npz", **g)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present: golden fixtures can only be regenerated in the build container")
    main()
