#!/usr/bin/env python3
"""Golden GRADIENTS of the physRNN "Hidden" model from the shipped TorchScript artefact itself
(rnn/saved_models/physRNN-Hidden_lr0.0007.neur128-128_xv4_mp1_num14564_BEST_script_cpu.pt, torch.jit.load on the CPU): torch
autograd through the serialised graph -- what the reference's training loop differentiates (train_rnn_rollout_torchscript_hydra.py
builds the same class and scripts it).  Loss = <out, u0> + <out_sfc, u1> + <mem_out, u2> with seeded upstream gradients u, so the
stored gradients are the vector-Jacobian products the HIP backward (csa_phys_train_backward) must reproduce.
Stores: seeds, the artefact's hx2 draw, upstream gradients, d(loss)/d(every parameter), d(loss)/d(rnn_mem).  Data only."""
import os
import numpy as np
import torch
from make_golden_physrnn import ART, OUT, inputs


def upstream(B, seed):
    g = torch.Generator().manual_seed(900 + seed)
    return torch.randn(B, 60, 5, generator=g), torch.randn(B, 8, generator=g), 0.2 * torch.randn(B, 50, 16, generator=g)


def main(cases=((8, 11), (37, 12))):
    m = torch.jit.load(ART, map_location="cpu").train()
    P = {k: v.detach().float() for k, v in m.state_dict().items()}
    d = {}
    for i, (B, seed) in enumerate(cases):
        xm, xs, mem, xd = inputs(P, B, seed)
        mem = mem.clone().requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        torch.manual_seed(1000 + seed)
        out, out_sfc, mem_out = m([xm.clone(), xs.clone(), mem, xd.clone()])
        torch.manual_seed(1000 + seed)
        hx2 = torch.randn(B, 128)
        u = upstream(B, seed)
        ((out * u[0]).sum() + (out_sfc * u[1]).sum() + (mem_out * u[2]).sum()).backward()
        d[f"case{i}.cfg"] = np.array([B, seed], np.int64)
        d[f"case{i}.hx2"] = hx2.numpy()
        for k, v in (("out", out), ("out_sfc", out_sfc), ("mem_out", mem_out)):
            d[f"case{i}.{k}"] = v.detach().numpy()
        d[f"case{i}.g.rnn_mem"] = mem.grad.numpy()
        n = 0
        for k, p in m.named_parameters():
            if p.grad is not None:
                d[f"case{i}.g.{k}"] = p.grad.numpy().copy(); n += 1
        print(f"case {i}: B={B}, {n} parameter gradients, |d mem| max {mem.grad.abs().max():.3e}")
    np.savez_compressed(os.path.join(OUT, "physrnn_hidden_grads.npz"), **d)


if __name__ == "__main__":
    main()
