import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import torch_ref
from test_cnn_baseline import _arch
from climsim_amd.baselines import CNNTrainer
for depth, width, B, dropout in [(12, 406, 6, 0.0), (12, 406, 6, 0.0), (6, 406, 6, 0.0), (12, 400, 6, 0.0), (12, 200, 6, 0.0)]:
    ws, bs = _arch(depth, width, seed=3)
    tr = CNNTrainer([w.numpy() for w in ws], [b.numpy() for b in bs], depth=depth, width=width, dropout=dropout, max_batch=8)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 60, 6, generator=g); yt = torch.randn(B, 60, 10, generator=g)
    y = tr.forward(x.cuda(), None).cpu()
    loss, grads = tr.backward(yt.cuda())
    dt = torch.float32
    wd = [w.to(dt).requires_grad_(True) for w in ws]; bd = [b.to(dt).requires_grad_(True) for b in bs]
    yr = torch_ref.cnn_ref(x.to(dt), wd, bd, depth=depth, dropout=dropout)
    torch_ref.mae_adjusted(yt.to(dt), yr).backward()
    gw, gb = tr.unpack(grads)
    ew = [((a.double() - r.grad.double()).abs().max() / r.grad.double().abs().max()).item() for a, r in zip(gw, wd)]
    eb = [((a.double() - r.grad.double()).abs().max() / r.grad.double().abs().max()).item() for a, r in zip(gb, bd)]
    print(depth, width, "fwd", ((y - yr.detach()).abs().max() / yr.abs().max()).item(), "n sign-sensitive", int(((y - yt).abs() < 1e-5).sum()),
          "relu-out near 0:", int(((yr[:, :, 2:] > 0) & (yr[:, :, 2:] < 1e-6)).sum()))
    print("   w:", " ".join(f"{e:.0e}" for e in ew))
    print("   b:", " ".join(f"{e:.0e}" for e in eb))
