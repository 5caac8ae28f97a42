#!/usr/bin/env python3
"""Golden loss scalars and gradients of a T_w = 3 TBPTT window for mp_mode -1 (the model predicts the liquid fraction itself,
rnn/models/models.py:303-329) and mp_mode -2 (it also predicts the total-water tendency and the cloud fraction of total water,
:286-301): the reference's own RNN_autoreg + rnn/metrics.py, the loss assembly of rnn/utils.py:1203-1335 (huber + 6e-6 energy +
6e7 water), as make_golden_current.py does for mp_mode 1.  Weights: the cur_mpm1 / cur_mpm2 models of make_golden_variants.py.
Build container only; data-only output (cur_mpm1_train.npz, cur_mpm2_train.npz)."""
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import make_golden_current as G  # noqa: E402
import make_golden_variants as V  # noqa: E402
from synth import synth_inputs  # noqa: E402

torch.set_num_threads(4)


def main(tag="cur_mpm1"):
    ref_models, ref_metrics = G.import_reference()
    c = G.consts()
    d0 = np.load(f"{OUT}/{tag}_model.npz")
    coeffs = {k: c[k] for k in ("xmean_lev", "xmean_sca", "xdiv_lev", "xdiv_sca", "hyai", "hybi", "hyam", "hybm", "lbd_qc", "lbd_qi", "lbd_qn")}
    coeffs["yscale_lev"], coeffs["yscale_sca"] = d0["c.yscale_lev"], d0["c.yscale_sca"]
    coeffs["xmean_lev"], coeffs["xdiv_lev"] = d0["c.xmean_lev"], d0["c.xdiv_lev"]          # (60, 16) for cur_mpm2
    cfg = V.cfg_for(tag)                             # ny 6, nneur (64, 64), output_prune, mp_mode -1 / -2
    torch.manual_seed(0)
    model = ref_models.RNN_autoreg(cfg, coeffs, torch.device("cpu"))
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(d0["w." + k]))
    model.train()
    B, T_w = 5, 3
    io = {"grad.B": np.array(B, np.int32), "grad.T_w": np.array(T_w, np.int32)}
    mem0 = (0.1 * torch.randn(60, B, 16)).requires_grad_(True)
    mem, outs, outs_sfc, xraws, sps = mem0, [], [], [], []
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from oracle import torch_ref
    c15 = {k: (v[:, :15] if k in ("xmean_lev", "xdiv_lev") else v) for k, v in c.items()}
    for t in range(T_w):
        x_main, x_sfc = synth_inputs(c15, B, 5100 + t)
        xm, xs = torch.from_numpy(x_main), torch.from_numpy(x_sfc)
        if cfg.nx == 16:                             # specific humidity as the last input column (make_golden_variants.py)
            pres = model.hyam * 100000.0 + xs[:, 0:1] * model.hybm
            xm = torch.cat((xm, torch_ref.EmulatorRef.rh_to_q(torch_ref.EmulatorRef, xm[:, :, 1], xm[:, :, 0], pres).unsqueeze(2)), 2)
            x_main = xm.numpy().copy()
        xn = xm.clone()
        xn[:, :, 2] = 1 - torch.exp(-xn[:, :, 2] * model.lbd_qc)
        xn[:, :, 3] = 1 - torch.exp(-xn[:, :, 3] * model.lbd_qi)
        xn = torch.nan_to_num((xn - model.xmean_lev) / model.xdiv_lev, 0.0, 0.0, 0.0)
        xsn = (xs - model.xmean_sca) / model.xdiv_sca
        out, out_sfc, mem = model([xn, xsn, mem])
        outs.append(out); outs_sfc.append(out_sfc); xraws.append(xm)
        sps.append(xsn[:, 0:1] * model.xdiv_sca[0:1] + model.xmean_sca[0:1])
        io[f"grad.t{t}.x_main"], io[f"grad.t{t}.x_sfc"] = x_main, x_sfc
    preds, preds_sfc = torch.cat(outs, 0), torch.cat(outs_sfc, 0)
    g = np.random.Generator(np.random.PCG64(778))
    tgt = torch.from_numpy(g.standard_normal(preds.shape).astype(np.float32))
    tgt[:, :, 3] = torch.from_numpy(g.uniform(0, 1, preds.shape[:2]).astype(np.float32)) * model.yscale_lev[:, 3]   # a liquid fraction in [0, 1]
    if tag == "cur_mpm2":                            # column 2: (cloud fraction of total water)^(1/4); cloud water is ~1e-4..1e-2 of the total
        tgt[:, :, 2] = torch.from_numpy(g.uniform(0.1, 0.35, preds.shape[:2]).astype(np.float32)) * model.yscale_lev[:, 2]
        tgt[:, :, 1] *= 0.05                         # total-water tendency: a few percent of the normalisation scale
    tgt_sfc = torch.from_numpy(g.standard_normal(preds_sfc.shape).astype(np.float32))
    huber, mse, mae = ref_metrics.metrics_flatten(tgt, tgt_sfc, preds, preds_sfc)
    x_raw, sp = torch.cat(xraws, 0), torch.cat(sps, 0)
    ypo, ypo_sfc = model.postprocessing(preds, preds_sfc, x_raw)
    yto, yto_sfc = model.postprocessing(tgt, tgt_sfc, x_raw)
    em = ref_metrics.get_energy_metric(c["hyai"], c["hybi"], "cpu")
    wc = ref_metrics.get_water_conservation(c["hyai"], c["hybi"], "cpu")
    h_con = em(yto, yto_sfc, ypo, ypo_sfc, sp, T_w)
    w_con = torch.mean(torch.square(wc(ypo, ypo_sfc, sp, None, x_raw, 1) - wc(yto, yto_sfc, sp, None, x_raw, 1)))
    precip = ref_metrics.precip_sum_mse(yto_sfc, ypo_sfc, T_w)
    loss = torch.stack([huber, 6e-6 * h_con, 6e7 * w_con]).sum()
    loss.backward()
    io["grad.mem0"], io["grad.tgt"], io["grad.tgt_sfc"] = mem0.detach().numpy(), tgt.numpy(), tgt_sfc.numpy()
    io["grad.yto"], io["grad.yto_sfc"] = yto.detach().numpy(), yto_sfc.detach().numpy()
    io["grad.preds"], io["grad.mem_final"] = preds.detach().numpy(), mem.detach().numpy()
    for name, val in (("huber", huber), ("mse", mse), ("mae", mae), ("energy", h_con), ("water", w_con), ("precip_sum_mse", precip), ("loss", loss)):
        io["grad.loss." + name] = np.array(val.item(), np.float64)
    io["grad.d_mem0"] = mem0.grad.numpy().copy()
    for k, p in model.named_parameters():
        if p.grad is None:
            print("no gradient reaches", k)
        io["grad.dw." + k] = (torch.zeros_like(p) if p.grad is None else p.grad).numpy().copy()
    print({k[10:]: float(v) for k, v in io.items() if k.startswith("grad.loss.")})
    np.savez_compressed(f"{OUT}/{tag}_train.npz", **io)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main(sys.argv[1])
        sys.exit(0)
    if not os.path.isdir(G.REF):
        sys.exit("reference not present: golden fixtures can only be regenerated in the build container")
    main()
