#!/usr/bin/env python3
"""Golden vectors from the reference's CURRENT model class rnn/models/models.py::RNN_autoreg
and its loss functions rnn/metrics.py, imported from /root/reference in the build container.

Import notes: models.py imports omegaconf only for type annotations and
models_torch_kernels.py compiles an inline CUDA extension at import time (no nvcc here, and
the CPU branch never touches it), so the two names are neutralised before the import.  No
reference source or bytecode is written anywhere: outputs are fp32 arrays in .npz files.

No trained checkpoint of the current generation ships with the reference (rnn/saved_models/
holds physRNN models only), so weights are the class's own seeded random initialisation,
exported next to the I/O.  Constants come from the shipped artefacts' buffers.
"""
import glob
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/rnn"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from synth import synth_inputs  # noqa: E402

torch.set_num_threads(4)


def import_reference():
    stub = types.ModuleType("omegaconf")
    stub.DictConfig = dict
    stub.OmegaConf = object
    sys.modules["omegaconf"] = stub
    import torch.utils.cpp_extension as ce
    ce.load_inline = lambda *a, **k: None
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "models"))
    import models as ref_models  # rnn/models/models.py via rnn/models/__init__? fall through below
    if not hasattr(ref_models, "RNN_autoreg"):
        from models import models as ref_models
    import metrics as ref_metrics
    return ref_models, ref_metrics


def consts():
    d = np.load(f"{OUT}/v4_memory_model.npz")
    c = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
    f = sorted(glob.glob(f"{REF}/saved_models/*num14564*BEST_script_cpu.pt"))[0]
    m = torch.jit.load(f, map_location="cpu")
    bufs = dict(m.named_buffers())
    # the legacy wrappers carry hyam/hybm rounded to 5 digits; the physRNN artefact has the
    # full-precision grid (consistent with its hyai/hybi), which the current generation uses
    for k in ("hyai", "hybi", "hyam", "hybm", "lbd_qn"):
        c[k] = bufs[k].numpy().astype(np.float32).copy()
    return c


def make_cfg(use_lstm, nneur, output_prune, mp_mode=1):
    return types.SimpleNamespace(
        ny=5 if mp_mode == 1 else 6, nlev=60, nx=15, nx_sfc=19, ny_sfc=8, nneur=nneur, nh_mem=16,
        use_initial_mlp=True, add_pres=True, output_prune=output_prune, use_lstm=use_lstm,
        add_stochastic_layer=False, ensemble_size=1, mp_mode=mp_mode, separate_radiation=False)


def main():
    ref_models, ref_metrics = import_reference()
    c = consts()
    np.savez(f"{OUT}/grid_consts.npz", hyai=c["hyai"], hybi=c["hybi"], lbd_qn=c["lbd_qn"])
    coeffs = {k: c[k] for k in ("yscale_lev", "yscale_sca", "xmean_lev", "xmean_sca", "xdiv_lev",
                                "xdiv_sca", "hyai", "hybi", "hyam", "hybm", "lbd_qc", "lbd_qi", "lbd_qn")}
    variants = {
        "cur_lstm128": dict(use_lstm=True, nneur=(128, 128), output_prune=True),
        "cur_lstm144": dict(use_lstm=True, nneur=(144, 144), output_prune=True),
        "cur_gru128": dict(use_lstm=False, nneur=(128, 128), output_prune=False),
    }
    for tag, v in variants.items():
        torch.manual_seed({"cur_lstm128": 101, "cur_lstm144": 102, "cur_gru128": 103}[tag])
        cfg = make_cfg(**v)
        model = ref_models.RNN_autoreg(cfg, coeffs, torch.device("cpu")).eval()
        sd = {k: p.detach().numpy().astype(np.float32) for k, p in model.named_parameters()}
        d = {"c." + k: val for k, val in c.items() if k in (
            "xmean_lev", "xdiv_lev", "xmean_sca", "xdiv_sca", "lbd_qc", "lbd_qi", "yscale_lev",
            "yscale_sca", "hyam", "hybm")}
        d.update({"w." + k: val for k, val in sd.items()})
        d["flags.use_lstm"] = np.array(int(v["use_lstm"]), np.int32)
        d["flags.output_prune"] = np.array(int(v["output_prune"]), np.int32)
        np.savez(f"{OUT}/{tag}_model.npz", **d)
        print(tag, "params", sum(val.size for val in sd.values()))

        io = {}
        for B, seed in ((2, 31), (16, 32)):
            nsteps = 3
            mem = torch.zeros(60, B, 16)
            io[f"B{B}.nsteps"] = np.array(nsteps, np.int32)
            for t in range(nsteps):
                x_main, x_sfc = synth_inputs(c, B, seed * 100 + t)
                xm, xs = torch.from_numpy(x_main), torch.from_numpy(x_sfc)
                # the wrapper's own v4 preprocessing arithmetic (rnn/utils.py:200-217), done with
                # torch ops here because rnn/utils.py itself is not importable (numba/h5py absent)
                xn = xm.clone()
                xn[:, :, 2] = 1 - torch.exp(-xn[:, :, 2] * model.lbd_qc)
                xn[:, :, 3] = 1 - torch.exp(-xn[:, :, 3] * model.lbd_qi)
                xn = (xn - model.xmean_lev) / model.xdiv_lev
                xsn = (xs - model.xmean_sca) / model.xdiv_sca
                xn = torch.where(torch.isnan(xn), torch.tensor(0.0), xn)
                xn = torch.where(torch.isinf(xn), torch.tensor(0.0), xn)
                with torch.no_grad():
                    out, out_sfc, mem_out = model([xn, xsn, mem])
                    o6, osd = model.postprocessing(out.clone(), out_sfc.clone(), xm)
                p = f"B{B}.t{t}."
                io[p + "x_main"] = x_main
                io[p + "x_sfc"] = x_sfc
                io[p + "x_main_n"] = xn.numpy()
                io[p + "x_sfc_n"] = xsn.numpy()
                io[p + "mem_in"] = mem.numpy().copy()
                io[p + "out"] = out.numpy().copy()
                io[p + "out_sfc"] = out_sfc.numpy().copy()
                io[p + "mem_out"] = mem_out.numpy().copy()
                io[p + "post_lev"] = o6.numpy().copy()
                io[p + "post_sfc"] = osd.numpy().copy()
                mem = mem_out.detach().clone()
                print(tag, B, t, "finite", bool(torch.isfinite(o6).all()), float(out.abs().max()))

        # ---- gradients through a T_w = 3 TBPTT window (config 3) ----------------
        if True:
            B, T_w = 6, 3
            model.train()
            model.zero_grad()
            mem0 = (0.1 * torch.randn(60, B, 16)).requires_grad_(True)
            mem = mem0
            outs, outs_sfc, xraws, sps = [], [], [], []
            for t in range(T_w):
                x_main, x_sfc = synth_inputs(c, B, 4100 + t)
                xm, xs = torch.from_numpy(x_main), torch.from_numpy(x_sfc)
                xn = xm.clone()
                xn[:, :, 2] = 1 - torch.exp(-xn[:, :, 2] * model.lbd_qc)
                xn[:, :, 3] = 1 - torch.exp(-xn[:, :, 3] * model.lbd_qi)
                xn = (xn - model.xmean_lev) / model.xdiv_lev
                xsn = (xs - model.xmean_sca) / model.xdiv_sca
                xn = torch.where(torch.isnan(xn), torch.tensor(0.0), xn)
                xn = torch.where(torch.isinf(xn), torch.tensor(0.0), xn)
                out, out_sfc, mem = model([xn, xsn, mem])
                outs.append(out); outs_sfc.append(out_sfc); xraws.append(xm)
                # utils.py:1248: surface pressure de-normalised from the normalised x_sfc
                sps.append(xsn[:, 0:1] * model.xdiv_sca[0:1] + model.xmean_sca[0:1])
                io[f"grad.t{t}.x_main"] = x_main
                io[f"grad.t{t}.x_sfc"] = x_sfc
            preds = torch.cat(outs, 0); preds_sfc = torch.cat(outs_sfc, 0)
            g = np.random.Generator(np.random.PCG64(777))
            tgt = torch.from_numpy(g.standard_normal(preds.shape).astype(np.float32))
            tgt_sfc = torch.from_numpy(g.standard_normal(preds_sfc.shape).astype(np.float32))
            # loss assembly of rnn/utils.py:1203-1335 with the default weights of
            # rnn/conf/autoreg_LSTM_longwindows.yaml:60-66 (huber + 6e-6*energy + 6e7*water)
            huber, mse, mae = ref_metrics.metrics_flatten(tgt, tgt_sfc, preds, preds_sfc)
            x_raw = torch.cat(xraws, 0); sp = torch.cat(sps, 0)
            ypo, ypo_sfc = model.postprocessing(preds, preds_sfc, x_raw)
            yto, yto_sfc = model.postprocessing(tgt, tgt_sfc, x_raw)
            em = ref_metrics.get_energy_metric(c["hyai"], c["hybi"], "cpu")
            wc = ref_metrics.get_water_conservation(c["hyai"], c["hybi"], "cpu")
            h_con = em(yto, yto_sfc, ypo, ypo_sfc, sp, T_w)
            # utils.py:1257-1259: per-sample water closure of prediction vs target, timesteps=1
            wcon_p = wc(ypo, ypo_sfc, sp, None, x_raw, 1)
            wcon_t = wc(yto, yto_sfc, sp, None, x_raw, 1)
            w_con = torch.mean(torch.square(wcon_p - wcon_t))
            precip = ref_metrics.precip_sum_mse(yto_sfc, ypo_sfc, T_w)
            loss = torch.stack([huber, 6e-6 * h_con, 6e7 * w_con]).sum()
            loss.backward()
            io["grad.B"] = np.array(B, np.int32); io["grad.T_w"] = np.array(T_w, np.int32)
            io["grad.mem0"] = mem0.detach().numpy().copy()
            io["grad.tgt"] = tgt.numpy(); io["grad.tgt_sfc"] = tgt_sfc.numpy()
            io["grad.preds"] = preds.detach().numpy().copy()
            io["grad.preds_sfc"] = preds_sfc.detach().numpy().copy()
            io["grad.mem_final"] = mem.detach().numpy().copy()
            for name, val in (("huber", huber), ("mse", mse), ("mae", mae), ("energy", h_con),
                              ("water", w_con), ("precip_sum_mse", precip), ("loss", loss)):
                io["grad.loss." + name] = np.array(val.item(), np.float64)
            io["grad.d_mem0"] = mem0.grad.numpy().copy()
            for k, p in model.named_parameters():
                io["grad.dw." + k] = p.grad.numpy().copy()
            print(tag, "loss", loss.item(), "huber", huber.item(), "energy", h_con.item(), "water", w_con.item(),
                  "precip", precip.item())
            model.eval()
        np.savez_compressed(f"{OUT}/{tag}_io.npz", **io)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present: golden fixtures can only be regenerated in the build container")
    main()
