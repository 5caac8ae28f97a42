#!/usr/bin/env python3
"""Golden gradients of ensemble / CRPS training of the stochastic model (SURVEY section 8 rows a9 + f3): the reference's own
rnn/models/models.py::RNN_autoreg(add_stochastic_layer=True) and rnn/metrics.py::CRPS, autograd over a T_w = 2 window with
E = 2 members per column, as the training loop builds it (rnn/utils.py:1065-1075 member-major replication of the inputs,
:1098-1137 forward with the memory fed back, :1213 the score as loss).  The model draws hx0, cx0 = randn(E*B, nh) and
eps = randn(nlev, E*B, nh) inside forward; the draws are reproduced by re-seeding and stored so the HIP trainer replays them.
Build container only; data-only outputs."""
import os
import sys

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import make_golden_current as G  # noqa: E402
from synth import synth_inputs  # noqa: E402

torch.set_num_threads(4)


def main():
    ref_models, ref_metrics = G.import_reference()
    c = G.consts()
    coeffs = {k: c[k] for k in ("yscale_lev", "yscale_sca", "xmean_lev", "xmean_sca", "xdiv_lev", "xdiv_sca", "hyai", "hybi", "hyam",
                                "hybm", "lbd_qc", "lbd_qi", "lbd_qn")}
    d0 = np.load(f"{OUT}/cur_stoch_model.npz")             # the stochastic model of make_golden_variants.py: same weights
    cfg = G.make_cfg(use_lstm=True, nneur=(128, 128), output_prune=bool(d0["flags.output_prune"]))
    cfg.add_stochastic_layer = True
    torch.manual_seed(0)
    model = ref_models.RNN_autoreg(cfg, coeffs, torch.device("cpu"))
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(d0["w." + k]))
    model.train()
    B, E, Tw = 3, 2, 2
    BE = B * E
    io = {"B": np.array(B), "E": np.array(E), "T_w": np.array(Tw)}
    mem0 = (0.1 * torch.randn(60, BE, 16)).requires_grad_(True)
    mem, outs, outs_sfc = mem0, [], []
    for t in range(Tw):
        x_main, x_sfc = synth_inputs(c, B, 8100 + t)
        xm, xs = torch.from_numpy(x_main), torch.from_numpy(x_sfc)
        xn = xm.clone()
        xn[:, :, 2] = 1 - torch.exp(-xn[:, :, 2] * model.lbd_qc)
        xn[:, :, 3] = 1 - torch.exp(-xn[:, :, 3] * model.lbd_qi)
        xn = torch.nan_to_num((xn - model.xmean_lev) / model.xdiv_lev, 0.0, 0.0, 0.0)
        xsn = (xs - model.xmean_sca) / model.xdiv_sca
        # rnn/utils.py:1065-1069: member-major replication
        xe = torch.repeat_interleave(xn.unsqueeze(0), repeats=E, dim=0).flatten(0, 1)
        xse = torch.repeat_interleave(xsn.unsqueeze(0), repeats=E, dim=0).flatten(0, 1)
        seed = 7700 + t
        torch.manual_seed(seed)
        out, out_sfc, mem = model([xe, xse, mem])
        torch.manual_seed(seed)
        io[f"t{t}.hx0"], io[f"t{t}.cx0"] = torch.randn(BE, 128).numpy(), torch.randn(BE, 128).numpy()
        io[f"t{t}.eps"] = torch.randn(60, BE, 128).numpy()
        io[f"t{t}.x_main_n"], io[f"t{t}.x_sfc_n"] = xn.numpy(), xsn.numpy()
        outs.append(out); outs_sfc.append(out_sfc)
    preds, preds_sfc = torch.cat(outs, 0), torch.cat(outs_sfc, 0)
    g = np.random.Generator(np.random.PCG64(4242))
    tgt = torch.from_numpy((0.5 * g.standard_normal((Tw * B, 60, 5))).astype(np.float32))
    tgt_sfc = torch.from_numpy((0.5 * g.standard_normal((Tw * B, 8))).astype(np.float32))
    loss = ref_metrics.CRPS(tgt, tgt_sfc, preds, preds_sfc, Tw, beta=1, alpha=1.0)
    loss.backward()
    io["mem0"], io["tgt"], io["tgt_sfc"] = mem0.detach().numpy(), tgt.numpy(), tgt_sfc.numpy()
    io["preds"], io["preds_sfc"], io["mem_final"] = preds.detach().numpy(), preds_sfc.detach().numpy(), mem.detach().numpy()
    io["loss"] = np.array(loss.item(), np.float64)
    io["d_mem0"] = mem0.grad.numpy().copy()
    for k, p in model.named_parameters():
        io["dw." + k] = p.grad.numpy().copy()
    print("loss", loss.item(), "max|dw|", max(float(p.grad.abs().max()) for p in model.parameters()))
    np.savez_compressed(f"{OUT}/cur_stoch_train.npz", **io)


if __name__ == "__main__":
    if not os.path.isdir(G.REF):
        sys.exit("reference not present: golden fixtures can only be regenerated in the build container")
    main()
