#!/usr/bin/env python3
"""Golden vectors from the reference class rnn/models/models.py::RNN_autoreg for the variants the main
generator (make_golden_current.py) does not cover:
  cur_mpm1   mp_mode -1  (6 outputs, predicted liquid fraction; postprocessing models.py:310-337)
  cur_mpm2   mp_mode -2  (total water + cloud fraction; models.py:286-301), 16 level inputs with specific
             humidity appended as the last one (rnn/utils.py:262-271; q itself is test-side input data here)
  cur_stoch  add_stochastic_layer=True, LSTM: rnn0 (down, randn init) -> rnn1 (up) -> MyStochasticLSTMLayer4
             (down) (models.py:405-412,464-474,521-534).  The three randn draws of a forward (hx0, cx0, eps)
             are reproduced under the same seed and stored.
Weights are the class's seeded random initialisation (no trained checkpoints of these variants ship).  Run with
TORCHDYNAMO_DISABLE=1 (set below).  Build container only; outputs are fp32 arrays, no reference code."""
import os
import sys
import types
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
import make_golden_current as G  # noqa: E402
from synth import synth_inputs  # noqa: E402
from oracle import torch_ref  # noqa: E402  (only for the test-side q input of cur_mpm2)


def cfg_for(tag):
    base = dict(nlev=60, nx_sfc=19, ny_sfc=8, nh_mem=16, use_initial_mlp=True, add_pres=True, ensemble_size=1,
                separate_radiation=False)
    if tag == "cur_mpm1":
        base.update(ny=6, nx=15, nneur=(64, 64), output_prune=True, use_lstm=True, add_stochastic_layer=False, mp_mode=-1)
    elif tag == "cur_mpm2":
        base.update(ny=6, nx=16, nneur=(64, 64), output_prune=False, use_lstm=True, add_stochastic_layer=False, mp_mode=-2)
    else:
        base.update(ny=5, nx=15, nneur=(128, 128), output_prune=True, use_lstm=True, add_stochastic_layer=True, mp_mode=1)
    return types.SimpleNamespace(**base)


def main():
    ref_models, _ = G.import_reference()
    c0 = G.consts()
    for tag, seed in (("cur_mpm1", 201), ("cur_mpm2", 202), ("cur_stoch", 203)):
        c = dict(c0)
        cfg = cfg_for(tag)
        if cfg.ny == 6:   # the shipped yscale is (60,5): the extra output (a fraction) gets scale 1
            ys = c["yscale_lev"]
            one = np.ones((60, 1), np.float32)
            c["yscale_lev"] = (np.concatenate([ys[:, :3], one, ys[:, 3:]], 1) if tag == "cur_mpm1"
                               else np.concatenate([ys[:, :2], one, one, ys[:, 3:]], 1)).astype(np.float32)
        if cfg.nx == 16:
            qmean = np.geomspace(2e-6, 8e-3, 60).astype(np.float32)[:, None]
            c["xmean_lev"] = np.concatenate([c["xmean_lev"], qmean], 1)
            c["xdiv_lev"] = np.concatenate([c["xdiv_lev"], 4 * qmean], 1)
        coeffs = {k: c[k] for k in ("yscale_lev", "yscale_sca", "xmean_lev", "xmean_sca", "xdiv_lev", "xdiv_sca",
                                    "hyai", "hybi", "hyam", "hybm", "lbd_qc", "lbd_qi", "lbd_qn")}
        torch.manual_seed(seed)
        model = ref_models.RNN_autoreg(cfg, coeffs, torch.device("cpu")).eval()
        sd = {k: p.detach().numpy().astype(np.float32) for k, p in model.named_parameters()}
        d = {"c." + k: c[k] for k in ("xmean_lev", "xdiv_lev", "xmean_sca", "xdiv_sca", "lbd_qc", "lbd_qi",
                                     "yscale_lev", "yscale_sca", "hyam", "hybm")}
        d.update({"w." + k: v for k, v in sd.items()})
        d["flags.use_lstm"] = np.array(1, np.int32)
        d["flags.output_prune"] = np.array(int(cfg.output_prune), np.int32)
        d["flags.mp_mode"] = np.array(cfg.mp_mode, np.int32)
        d["flags.add_stochastic_layer"] = np.array(int(cfg.add_stochastic_layer), np.int32)
        np.savez(f"{OUT}/{tag}_model.npz", **d)
        print(tag, {k: v.shape for k, v in sd.items()})

        io = {}
        c15 = {k: (v[:, :15] if k in ("xmean_lev", "xdiv_lev") else v) for k, v in c.items()}
        qhelper = lambda rh, T, p: torch_ref.EmulatorRef.rh_to_q(torch_ref.EmulatorRef, rh, T, p)
        for B, s in ((3, 51), (10, 52)):
            nsteps = 2
            io[f"B{B}.nsteps"] = np.array(nsteps, np.int32)
            mem = torch.zeros(60, B, 16)
            for t in range(nsteps):
                x_main, x_sfc = synth_inputs(c15, B, s * 100 + t)
                xm, xs = torch.from_numpy(x_main), torch.from_numpy(x_sfc)
                if cfg.nx == 16:
                    pres = model.hyam * 100000.0 + xs[:, 0:1] * model.hybm
                    q = qhelper(xm[:, :, 1], xm[:, :, 0], pres)
                    xm = torch.cat((xm, q.unsqueeze(2)), 2)
                xn = xm.clone()
                xn[:, :, 2] = 1 - torch.exp(-xn[:, :, 2] * model.lbd_qc)
                xn[:, :, 3] = 1 - torch.exp(-xn[:, :, 3] * model.lbd_qi)
                xn = (xn - model.xmean_lev) / model.xdiv_lev
                xsn = (xs - model.xmean_sca) / model.xdiv_sca
                xn = torch.where(torch.isnan(xn), torch.tensor(0.0), xn)
                xn = torch.where(torch.isinf(xn), torch.tensor(0.0), xn)
                p = f"B{B}.t{t}."
                fseed = 9000 + 10 * s + t
                torch.manual_seed(fseed)
                with torch.no_grad():
                    out, out_sfc, mem_out = model([xn, xsn, mem])
                    o6, osd = model.postprocessing(out.clone(), out_sfc.clone(), xm)
                if cfg.add_stochastic_layer:
                    torch.manual_seed(fseed)
                    io[p + "hx0"] = torch.randn(B, 128).numpy()
                    io[p + "cx0"] = torch.randn(B, 128).numpy()
                    io[p + "eps"] = torch.randn(60, B, 128).numpy()
                io[p + "x_main"] = xm.numpy().copy()          # 16 wide for cur_mpm2 (q last)
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
                print(tag, B, t, "finite", bool(torch.isfinite(o6).all()), float(out.abs().max()), float(o6.abs().max()))
        np.savez_compressed(f"{OUT}/{tag}_io.npz", **io)


if __name__ == "__main__":
    if not os.path.isdir(G.REF):
        sys.exit("reference not present: golden fixtures can only be regenerated in the build container")
    main()
