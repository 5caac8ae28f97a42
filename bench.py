#!/usr/bin/env python3
"""bench.py -- grid-columns/s of the emulator forward and column-timesteps/s of the TBPTT training step on N MI355X
GPUs (one process per GPU).

    python bench.py --gpus N --steps K --warmup W        (N > 1: this process only SPAWNS the N ranks -- before it has
                                                          made any GPU call -- as `python -m torch.distributed.run ...`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W     (externally launched form, same result)

Default invocation = ONE JSON line on rank 0 with both halves of BASELINE.json's metric:
  * headline (`value`): BASELINE.json configs[1], the v4 stateless wrapper (weights of rnn/v4_rnn_wrapper_constrained.pt,
    committed as data under tests/golden/), one 384-column batch of synthetic raw inputs per GPU per step, inputs
    resident in HBM.  Columns are independent, so ranks shard them with NO data-path collective (weak scaling).
  * `memory_wrapper`: the same measurement for the stateful v4 memory wrapper (configs[2]/[4]'s model, the north_star's
    target), rnn1_mem fed back by the caller every step.
  * `train`: one TBPTT optimiser step (window 3) of the current-generation LSTM-with-memory, 384 columns per GPU, ONE
    flat-gradient all-reduce per step; column-timesteps/s, with the roofline of the BPTT recurrence kernel.
`roofline` is for the dominant kernel (the register-stationary recurrent kernel, two launches per step), timed with HIP
events on the launch stream; `cpu_baseline` is the oracle's torch restatement (nn.LSTM on the host cores), same batch.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of a process that has not touched the
    GPU (no torch import yet, no HIP call) and pass their output through.  Never a re-exec."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def _early_spawn():
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    a, _ = ap.parse_known_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))


if __name__ == "__main__":
    _early_spawn()

import numpy as np      # noqa: E402  (after the spawn decision: the parent of an N-rank run stays torch-free)
import torch            # noqa: E402

PEAK_FP32_TFLOPS = 157.3   # MI355X dense fp32 (vector == f32 MFMA), MI355X_MICROARCH.md
WORKLOADS = {
    # name: (model npz, columns per GPU, FLOP per column, HBM bytes per column)
    "v4_stateless_384": ("v4_stateless", 384, 31.79e6, 5148),
    "v4_memory_384": ("v4_memory", 384, 32.95e6, 12828),
    "v4_memory_2700": ("v4_memory", 2700, 32.95e6, 12828),
    "v4_memory_48": ("v4_memory", 48, 32.95e6, 12828),      # the low-res grid strong-scaled over 8 GPUs: latency case
}


def workload_spec(name):
    """(model npz, columns per GPU, FLOP per column, HBM bytes per column) of a forward workload; v4_memory_<B> / v4_stateless_<B>
    exist at any column count (the strong-scaled shapes of --gpus N)."""
    if name in WORKLOADS:
        return WORKLOADS[name]
    tag, B = name.rsplit("_", 1)
    base = WORKLOADS[tag + "_384"]
    return (base[0], int(B), base[2], base[3])


def rec_kernel_name(B):
    """Which recurrent kernel api.hip launches for a call of B columns (defaults: csa_set_rec1_max_batch 256, CSA_REC4M from 544)."""
    if B <= 256:
        return "lstm_rec1_kernel<128>"
    return "lstm_rec4m_kernel<128>" if B >= 544 else "lstm_rec2_kernel<128,false,0>"


def load_model(tag):
    d = np.load(os.path.join(ROOT, "tests", "golden", f"{tag}_model.npz"))
    consts = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
    weights = {k[2:]: d[k] for k in d.files if k.startswith("w.")}
    return consts, weights


def pmc_traffic(workload, kernel_prefix="lstm_rec2_kernel"):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary
    (profiles/*_<workload>_pmc.json, produced by tools/profile_r1.sh + tools/summarize_profiles.py from
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command); None if absent."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_pmc.json")))
    if not fs:
        return None
    try:
        d = json.load(open(fs[-1]))
        for k, v in d["kernels"].items():
            if k.startswith(kernel_prefix):
                return v["hbm_bytes"]
    except Exception:
        return None
    return None


def cpu_baseline(tag, consts, weights, xm, xs, mem, hx, cx, budget_s=12.0):
    """The oracle's torch restatement (what the reference executes on CPU: ATen nn.LSTM/Linear)
    timed on this host's cores on the SAME batch; bounded to ~budget_s seconds."""
    from oracle import torch_ref
    ref = torch_ref.EmulatorRef(consts, weights, legacy=True)
    args = [torch.from_numpy(xm), torch.from_numpy(xs), None if mem is None else torch.from_numpy(mem),
            torch.from_numpy(hx), torch.from_numpy(cx)]
    # ATen's intra-op threading stops paying long before all host cores are in use on a 384-column
    # batch; give the CPU its best case: probe a few thread counts and keep the fastest.
    ncpu = os.cpu_count() or 1
    best = (None, 1e30)
    with torch.no_grad():
        for nt in sorted({min(ncpu, n) for n in (4, 8, 16, 32, 64, ncpu)}):
            torch.set_num_threads(nt)
            ref.wrapper_forward(*args)
            t0 = time.perf_counter()
            for _ in range(3):
                ref.wrapper_forward(*args)
            dt = (time.perf_counter() - t0) / 3
            if dt < best[1]:
                best = (nt, dt)
    cores = best[0]
    torch.set_num_threads(cores)
    budget_s = max(3.0, budget_s - 4.0)
    with torch.no_grad():
        for _ in range(2):
            ref.wrapper_forward(*args)
        t0 = time.perf_counter()
        n = 0
        while True:
            ref.wrapper_forward(*args)
            n += 1
            el = time.perf_counter() - t0
            if (el > budget_s and n >= 5) or n >= 400:
                break
    B = xm.shape[0]
    return {"value": B * n / el, "unit": "grid-columns/s", "cores": cores, "kind": "port",
            "sample": f"{n} forward calls of the same {B}-column batch through oracle/torch_ref.py "
                      f"(torch {torch.__version__} CPU, {cores} threads), {el:.1f} s"}


def _fence(dist):
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def _timed(step, steps, warmup, dist, preheat_s=0.25):
    """W untimed warm-up steps (plus an untimed pre-heat of the same step so that the timed region does not start on an
    idle-clocked device), then EXACTLY `steps` steps between barrier + synchronize fences; MAX over ranks."""
    from climsim_amd.sharding import max_over_ranks
    gc.collect(); gc.disable()     # a generation-2 collection over the interpreter's heap costs tens of ms (and lets the device
    for _ in range(warmup):        # clock down while the host is busy): done BEFORE the warm-up, never inside the timed steps
        step()
    torch.cuda.synchronize()
    t0, pre = time.perf_counter(), 0
    while time.perf_counter() - t0 < preheat_s:      # back-to-back launches (sustained load), one sync per chunk
        for _ in range(25):
            step()
        torch.cuda.synchronize()
        pre += 25
    _fence(dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    _fence(dist)
    el = time.perf_counter() - t0
    gc.enable()
    return max_over_ranks(el, device="cuda"), pre


def cpu_train_baseline(consts, weights, grid, B, Tw, args_cpu, budget_s=8.0):
    """CPU leg of the training metric: the oracle's torch restatement (nn.LSTM autograd + the metrics restatement + Adam),
    the loop semantics of rnn/train_rnn_rollout.py:762-933, on this host's cores; a few optimiser steps, bounded."""
    from oracle import torch_ref
    xn, xsn, xr, t5, t8, y6, y8 = args_cpu
    ref = torch_ref.EmulatorRef(consts, weights, legacy=False, use_lstm=True, output_prune=True, scrub_inf=True)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 16)
    torch.set_num_threads(cores)
    mem = torch.zeros(60, B, 16)
    cat = lambda t: torch.cat([t] * Tw, 0)

    def one(mem):
        outs, outs_sfc = [], []
        for _ in range(Tw):
            o, os_, mem = ref.model_forward(xn, xsn, mem)
            outs.append(o); outs_sfc.append(os_)
        loss, _ = torch_ref.window_loss(ref, torch.cat(outs, 0), torch.cat(outs_sfc, 0), cat(t5), cat(t8), cat(y6), cat(y8),
                                        cat(xr), cat(xsn), grid["hyai"], grid["hybi"], Tw)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return mem.detach()
    mem = one(mem)
    t0, n = time.perf_counter(), 0
    while True:
        mem = one(mem)
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s and n >= 2) or n >= 50:
            break
    return {"value": B * Tw * n / el, "unit": "column-timesteps/s", "cores": cores, "kind": "port",
            "sample": f"{n} TBPTT optimiser steps (window {Tw}, {B} columns) through oracle/torch_ref.py autograd "
                      f"(torch {torch.__version__} CPU, {cores} threads), {el:.1f} s"}


def train_leg(a, rank, world, dist, steps, warmup, with_cpu, B=384, tag="cur_lstm128"):
    """BASELINE.json configs[2]: one TBPTT optimiser step of the current-generation LSTM with memory (window T_w = 3, 384
    columns per GPU): 3 forwards with saved activations, loss, 3 backwards, ONE flat-buffer RCCL all-reduce, Adam.
    Unit: column-timesteps/s.  Returns the `train` object of the JSON line."""
    from climsim_amd.train import Trainer
    from synth import synth_inputs
    consts, weights = load_model(tag)
    grid = np.load(os.path.join(ROOT, "tests", "golden", "grid_consts.npz"))
    Tw = 3
    lstm = "lstm" in tag
    nh, G = weights["rnn1.weight_hh_l0"].shape[1], (4 if lstm else 3)
    tr = Trainer(consts, weights, grid["hyai"], grid["hybi"], use_lstm=lstm, output_prune=lstm, max_batch=B, max_window=Tw)
    xm, xs = synth_inputs(consts, B, 7000 + rank)
    g = torch.Generator().manual_seed(rank)
    xmt, xst = torch.from_numpy(xm), torch.from_numpy(xs)
    xn = xmt.clone()
    xn[:, :, 2] = 1 - torch.exp(-xn[:, :, 2] * torch.from_numpy(consts["lbd_qc"]))
    xn[:, :, 3] = 1 - torch.exp(-xn[:, :, 3] * torch.from_numpy(consts["lbd_qi"]))
    xn = torch.nan_to_num((xn - torch.from_numpy(consts["xmean_lev"])) / torch.from_numpy(consts["xdiv_lev"]), 0.0, 0.0, 0.0)
    xsn = (xst - torch.from_numpy(consts["xmean_sca"])) / torch.from_numpy(consts["xdiv_sca"])
    t5, t8 = torch.randn(B, 60, 5, generator=g), torch.randn(B, 8, generator=g)
    y6 = torch.randn(B, 60, 6, generator=g) * 1e-5
    y8 = torch.rand(B, 8, generator=g) * 1e-7
    dv = lambda t: [t.contiguous().cuda()] * Tw
    args = (dv(xn), dv(xsn), dv(xmt), dv(t5), dv(t8), dv(y6), dv(y8))
    state = {"mem": torch.zeros(60, B, 16, device="cuda"), "sc": None}

    def step():
        state["sc"], state["mem"], _ = tr.window_step(*args, state["mem"], world_size=world, global_columns=B * world)
    el, pre = _timed(step, steps, warmup, dist)
    # per-stage durations (HIP events on the launch stream), same step loop, outside the timed region
    tr.set_profiling(True)
    tr.reset_profile()
    for _ in range(20):
        step()
    prof = tr.get_profile()
    tr.set_profiling(False)
    out = None
    if rank == 0:
        bwd_ms, nb = prof["bwd_rec"]
        flop = B * 60 * 2.0 * G * nh * nh         # W_hh^T . d(gates): algorithmic FLOP of one BPTT recurrence launch
        ach = flop / (bwd_ms * 1e-3) / 1e12 if bwd_ms > 0 else 0.0
        value = world * B * Tw * steps / el
        # fwd + bwd ~ 3x forward FLOP per column-timestep (32.95 MFLOP forward for LSTM 128/128, nh_mem 16)
        flop_step = 3 * (60 * (2.0 * G * nh * (nh + 16) + 2.0 * G * nh * nh + 2 * 2.0 * G * nh * nh + 4096 + 2 * 16 * nh + 160) + 11776)
        out = {"metric": "train-step column-timesteps/sec (TBPTT window 3)", "value": value, "unit": "column-timesteps/s",
               "steps": steps, "warmup": warmup, "preheat_steps": pre, "ms_per_step": 1e3 * el / steps, "scaling": "weak", "dtype": "f32",
               "config": {"workload": ("" if tag == "cur_lstm128" else tag[4:] + "_") + f"train_tbptt3_{B}", "columns_per_gpu": B, "window": Tw,
                          "model": f"RNN_autoreg {'LSTM' if lstm else 'GRU'} {nh}/{nh}, nh_mem 16, mp_mode 1; huber + energy + water loss; Adam",
                          "parallelism": f"columns sharded x{world}, one flat-gradient all-reduce per step"},
               "roofline": {"bound": "mfma", "pipe": "v_pk_fma_f32 (fp32 vector peak = f32 MFMA peak, 157.3 TF: the two share the FMA lanes)",
                            "kernel": f"{'lstm' if lstm else 'gru'}_bwd_rec_kernel<{nh}> (one launch per RNN per backward, 6 per step)",
                            "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                            "traffic": pmc_traffic(f"train_tbptt3_{B}", "lstm_bwd_rec_kernel"),
                            "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/)",
                            "algorithmic_hbm_bytes_per_launch": B * 60 * (2 * 4 * nh + 2 * nh) * 4.0,
                            "flop_per_launch": flop, "avg_launch_ms": bwd_ms, "launches_timed": nb},
               "whole_path": {"flop_per_column_timestep": flop_step, "achieved_tflops": value / world * flop_step / 1e12,
                              "frac_fp32_peak": value / world * flop_step / 1e12 / PEAK_FP32_TFLOPS},
               "stage_ms": {k: v[0] for k, v in prof.items()}, "loss": state["sc"]["loss"]}
        if with_cpu:
            out["cpu_baseline"] = cpu_train_baseline(consts, weights, grid, B, Tw, (xn, xsn, xmt, t5, t8, y6, y8))
    tr.close()
    return out


def aux_workload(a, rank, world, dist, ret=False):
    """Secondary workloads (parity-test configurations, reported for DESIGN.md's tables; not the headline):
    cur_lstm144_384   current-generation tuple wrapper at the reference's default width nh = 144
    mlp_384 / cnn_384 offline Keras baselines (random weights of the reference architectures), forward."""
    import climsim_amd
    from climsim_amd.baselines import MLPBaseline, CNNBaseline
    from synth import synth_inputs
    B = int(a.workload.rsplit("_", 1)[1])        # columns per GPU are the workload's suffix
    g = torch.Generator().manual_seed(100 + rank)
    if a.workload.rsplit("_", 1)[0] in ("cur_lstm144", "cur_gru128", "cur_lstm128"):       # any column count: cur_gru128_2700 ...
        tag = a.workload.rsplit("_", 1)[0]
        consts, weights = load_model(tag)
        m = climsim_amd.model_wrapper(consts, weights, use_lstm="lstm" in tag, output_prune="lstm" in tag, max_batch=B)
        xm, xs = synth_inputs(consts, B, 9000 + rank)
        xs_ = [torch.from_numpy(xm).cuda(), torch.from_numpy(xs).cuda()]
        state = {"mem": torch.zeros(60, B, 16, device="cuda")}

        def step():
            o6, osfc, state["mem"] = m(xs_[0], xs_[1], state["mem"])
        nh, G = (144 if "144" in tag else 128), (4 if "lstm" in tag else 3)
        flop_col = 60 * (2.0 * G * nh * (nh + 16) + 2.0 * G * nh * nh + 2 * 2.0 * G * nh * nh + 4096 + 2 * 16 * nh + 160) + 13e3
        what = f"RNN_autoreg {'LSTM' if G == 4 else 'GRU'} {nh}/{nh} tuple wrapper"
    elif a.workload == "mlp_384":
        dims = [124, 768, 640, 512, 640, 640, 128]   # step2_retrain.py hidden widths (best HPO trial) + 128 outputs
        ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
        bs = [torch.zeros(dims[i + 1]) for i in range(len(dims) - 1)]
        m = MLPBaseline([w.numpy() for w in ws], [b.numpy() for b in bs], leaky_alpha=0.15, n_lin_out=120, max_batch=B)
        x = torch.randn(B, 124, generator=g).cuda()

        def step():
            m(x)
        flop_col, what = 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)), "Keras MLP baseline forward"
    elif a.workload.startswith("physrnn_e3sm_"):
        # the generation of the radiation scheme the frozen *_wrapped exports deploy (physics_rad_e3sm): weights of num94634
        import numpy as np
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden"))
        from make_golden_physrnn import inputs_rad
        from climsim_amd.physrnn import physical_RNN_autoreg
        gz = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "physrad16_e3sm.npz"))
        Pw = {k[2:]: torch.from_numpy(gz[k]) for k in gz.files if k.startswith("w.")}
        m = physical_RNN_autoreg(Pw, max_batch=B)
        xs_ = [t.cuda() for t in inputs_rad(Pw, B, 300 + rank)]
        hx2 = torch.randn(B, 128, generator=g).cuda()
        state = {"mem": xs_[2].transpose(0, 1).contiguous()}       # this generation takes and returns the memory level-major (50, B, 16)

        def step():
            o, osfc, state["mem"] = m([xs_[0], xs_[1], state["mem"], xs_[3]], hx2=hx2)
        # 50-level BiGRU + 11 x 16-wide head GEMM + 16-region decoder; radiation on 60 levels: LW gas optics 18-64-64-256 and
        # 2 x (128 -> 16); SW gas optics 2 models x 2 humidity variants x (7-32-32-112) and 2 x (112 -> 16); ~250 flop per
        # (level, g-point) cell of the cloud optics / two-stream / adding / LW sweeps
        flop_col = (50 * (2.0 * 20 * 128 + 2.0 * 384 * (143 + 128) + 2.0 * 384 * 256 + 2.0 * 212 * 128) + 50 * 16 * 150.0
                    + 60 * 2.0 * (18 * 64 + 64 * 64 + 64 * 256 + 2 * 128 * 16 + 4 * (7 * 32 + 32 * 32 + 32 * 112) + 2 * 112 * 16) + 60 * 16 * 250.0)
        what = "physRNN_physRad-16 nreg16, physics_rad_e3sm generation (BiGRU 128/128 over 50 levels + decoder + LW/SW gas-optics MLPs + solver), weights of num94634"
    elif a.workload.startswith("physrnn_wrapped_"):
        # the DEPLOYED module: a frozen `*_wrapped` export (raw inputs -> physical tendencies), weights of num13483 (variant a153783c)
        import numpy as np
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden"))
        from make_golden_frozen import inputs_wrapped
        from climsim_amd.physrnn import physical_RNN_wrapped
        gz = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "frozen_a153783c.npz"))
        Pw = {k[2:]: torch.from_numpy(gz[k]) for k in gz.files if k.startswith("w.")}
        FL = {k[5:]: int(gz[k]) for k in gz.files if k.startswith("flag.")}
        FL["band_idx"] = [int(v) for v in gz["cfg.band_idx"]]
        m = physical_RNN_wrapped(Pw, FL, max_batch=B)
        x0, s0, mem0 = (t.cuda() for t in inputs_wrapped(Pw, B, 300 + rank))
        hx2, mask_u = torch.randn(B, 128, generator=g).cuda(), torch.rand(60, B, 16, generator=g).cuda()
        state = {"mem": mem0}

        def step():   # the host's call: raw state in, tendencies out, memory fed back
            o, osfc, state["mem"] = m(x0, s0, state["mem"], hx2=hx2, mask_u=mask_u)
        flop_col = (50 * (2.0 * 20 * 128 + 2.0 * 384 * (143 + 128) + 2.0 * 384 * 256 + 2.0 * 192 * 128) + 50 * 16 * 150.0
                    + 60 * 2.0 * (18 * 64 + 64 * 64 + 64 * 256 + 2 * 128 * 16 + 4 * (7 * 32 + 32 * 32 + 32 * 16)) + 60 * 16 * 250.0)
        what = "frozen physRNN export (model_wrapper + nx21 physical_RNN_autoreg: BiGRU 128/128 over 50 levels, 16-region decoder, LW/SW gas optics, two-stream solver), weights of num13483"
    elif a.workload.startswith("physrnn_train_"):
        import numpy as np
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden"))
        from make_golden_physrnn import inputs as phys_inputs
        from climsim_amd.physrnn import physical_RNN_autoreg, physical_RNN_trainer
        gz = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "physrnn_hidden.npz"))
        Pw = {k[2:]: torch.from_numpy(gz[k]) for k in gz.files if k.startswith("w.")}
        m = physical_RNN_autoreg(Pw, max_batch=B)
        tr = physical_RNN_trainer(m)
        xs_ = [t.cuda() for t in phys_inputs(Pw, B, 300 + rank)]
        hx2 = torch.randn(B, 128, generator=g).cuda()
        tgt = [torch.randn(B, 60, 5, generator=g).cuda() * 0.1, torch.randn(B, 8, generator=g).abs().cuda() * 0.1]
        yto = [torch.randn(B, 60, 6, generator=g).cuda() * 1e-6, torch.randn(B, 8, generator=g).abs().cuda() * 1e-6]
        d_mem = torch.zeros(B, 50, 16, device="cuda")

        def step():   # forward, the reference trainer's loss (huber + energy + water) with its gradient, backward, all-reduce, Adam
            tr.zero_grad()
            o, osfc, _ = tr.forward(xs_, hx2=hx2)
            sc, d_o, d_s = tr.loss(o, osfc, tgt[0], tgt[1], yto[0], yto[1], xs_[3], xs_[1], scalars=False)
            tr.backward(d_o, d_s, d_mem)
            if world > 1:
                dist.all_reduce(tr.grads)
                tr.grads.mul_(1.0 / world)
            tr.adam_step(1e-5)
        flop_col = 3 * (60 * (2.0 * 22 * 128 + 2.0 * 384 * (143 + 128) + 2.0 * 384 * 256 + 2.0 * 192 * 128) + 50 * 16 * 150.0)
        what = ("physRNN-Hidden training step: forward with saved activations, huber + energy + water loss with its gradient, hand-written backward (decoder, BPTT through both GRUs, "
                "weight-gradient GEMMs), Adam + re-pack; fwd+bwd = 3x forward FLOP")
    elif a.workload.startswith("physrnn_") and not a.workload.startswith("physrnn_rad_"):
        import numpy as np
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden"))
        from make_golden_physrnn import inputs as phys_inputs
        from climsim_amd.physrnn import physical_RNN_autoreg
        gz = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "physrnn_hidden.npz"))
        Pw = {k[2:]: torch.from_numpy(gz[k]) for k in gz.files if k.startswith("w.")}
        m = physical_RNN_autoreg(Pw, max_batch=B)
        xs_ = [t.cuda() for t in phys_inputs(Pw, B, 300 + rank)]
        hx2 = torch.randn(B, 128, generator=g).cuda()
        state = {"mem": xs_[2]}

        def step():   # autoregressive use: the returned memory (latent + stored water) is fed back
            o, osfc, state["mem"] = m([xs_[0], xs_[1], state["mem"], xs_[3]], hx2=hx2)
        flop_col = 60 * (2.0 * 22 * 128 + 2.0 * 384 * (143 + 128) + 2.0 * 384 * 256 + 2.0 * 192 * 128) + 50 * 16 * 150.0
        what = "physRNN-Hidden (BiGRU 128/128 + microphysics decoder), weights of the shipped artefact"
    elif a.workload.startswith("physrnn_rad_"):
        import numpy as np
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden"))
        from make_golden_physrnn import inputs_rad
        from climsim_amd.physrnn import physical_RNN_autoreg
        gz = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "physrnn_rad.npz"))
        Pw = {k[2:]: torch.from_numpy(gz[k]) for k in gz.files if k.startswith("w.")}
        m = physical_RNN_autoreg(Pw, max_batch=B)
        xs_ = [t.cuda() for t in inputs_rad(Pw, B, 300 + rank)]
        hx2 = torch.randn(B, 128, generator=g).cuda()
        state = {"mem": xs_[2]}

        def step():
            o, osfc, state["mem"] = m([xs_[0], xs_[1], state["mem"], xs_[3]], hx2=hx2)
        # 50-level BiGRU + 59-wide head GEMM + decoder; radiation: gas optics 18-64-64-256, 2 x (128 -> 16), SW head 24-32-48
        # on 60 levels, ~250 flop per (level, g-point) cell of the two-stream / adding / LW sweeps
        flop_col = (50 * (2.0 * 20 * 128 + 2.0 * 384 * (143 + 128) + 2.0 * 384 * 256 + 2.0 * 60 * 128) + 50 * 4 * 150.0
                    + 60 * 2.0 * (18 * 64 + 64 * 64 + 64 * 256 + 2 * 128 * 16 + 24 * 32 + 32 * 48) + 60 * 16 * 250.0)
        what = "physRNN-Hidden radiation graph (BiGRU 128/128 over 50 levels + decoder + physical LW/SW scheme), weights of num4050"
    elif a.workload == "online_mlp_384":
        from climsim_amd.online import MLP, NewModel
        n_in, hidden = 557, [384, 1024, 640]              # v2_rh input vector, slurm/v2rh_mlp_*_3l_lr1em3.sbatch widths
        mm = MLP(n_in, 368, hidden, 3, output_prune=True, strato_lev_out=12)
        m = NewModel(mm, torch.zeros(n_in).numpy(), torch.ones(n_in).numpy(), torch.ones(368).numpy(),
                     torch.full((60,), 1e4).numpy(), torch.full((60,), 1e4).numpy(), max_batch=B)
        x = torch.randn(B, n_in, generator=g).cuda()

        def step():
            m(x)
        dims = [n_in] + hidden + [368]
        flop_col, what = 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)), "online MLP_v2rh wrapper forward (B, 557) -> (B, 368)"
    elif a.workload.startswith("cnn_train_"):    # 384: low-res grid; 512: the reference batch (hpo_train.py:294); 2700: configs[3] shard
        from climsim_amd.baselines import CNNTrainer
        depth, width = 12, 406
        ws, bs = [], []
        for i in range(depth):
            ci = 6 if i == 0 else width
            for co, cin, k in ((width, ci, 3), (width, width, 3), (width, ci, 1)):
                ws.append(torch.randn(co, cin, k, generator=g) * (0.9 / (cin * k) ** 0.5)); bs.append(torch.zeros(co))
        ws += [torch.randn(10, width, 1, generator=g) / width ** 0.5, torch.randn(10, 10, 1, generator=g) / 3]
        bs += [torch.zeros(10), torch.zeros(10)]
        tr = CNNTrainer([w.numpy() for w in ws], [b.numpy() for b in bs], max_batch=B)
        x = torch.randn(B, 60, 6, generator=g).cuda()
        yt = torch.randn(B, 60, 10, generator=g).cuda()

        def step():   # dropout masks drawn on the device, forward, mae_adjusted, backward, ONE flat all-reduce, Adam
            tr.train_step(x, yt, lr=1e-4, world_size=world)
        flop_col, what = 3 * 1.58e9, "Keras CNN training step (Dropout 0.175, mae_adjusted, Adam), fwd+bwd = 3x forward FLOP"
    else:
        depth, width = 12, 406
        ws, bs = [], []
        for i in range(depth):
            ci = 6 if i == 0 else width
            for co, cin, k in ((width, ci, 3), (width, width, 3), (width, ci, 1)):
                ws.append(torch.randn(co, cin, k, generator=g) * (0.9 / (cin * k) ** 0.5)); bs.append(torch.zeros(co))
        ws += [torch.randn(10, width, 1, generator=g) / width ** 0.5, torch.randn(10, 10, 1, generator=g) / 3]
        bs += [torch.zeros(10), torch.zeros(10)]
        m = CNNBaseline([w.numpy() for w in ws], [b.numpy() for b in bs], max_batch=B)
        x = torch.randn(B, 60, 6, generator=g).cuda()

        def step():
            m(x)
        flop_col, what = 1.58e9, "Keras CNN (12 residual Conv1D blocks) forward, implicit-GEMM"

    el, _ = _timed(step, a.steps, a.warmup, dist)
    if rank == 0:
        tf = B * flop_col * a.steps / el / 1e12
        out = ({
            "metric": "train-step columns/sec" if "train" in a.workload else "grid-columns/sec emulator fwd",
            "value": world * B * a.steps / el, "unit": "grid-columns/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": a.workload, "columns_per_gpu": B, "what": what,
                       "parallelism": f"columns sharded x{world}, " + ("one flat-gradient all-reduce per step" if "train" in a.workload else "no collective")},
            "whole_path": {"flop_per_column": flop_col, "achieved_tflops": tf, "frac_fp32_peak": tf / PEAK_FP32_TFLOPS}})
        if ret:
            return out
        print(json.dumps(out), flush=True)
    return None


AUX = ["cur_lstm144_384", "cur_lstm128_384", "cur_gru128_384", "mlp_384", "online_mlp_384", "physrnn_384", "physrnn_rad_384", "physrnn_2700", "physrnn_rad_2700", "physrnn_e3sm_384", "physrnn_e3sm_2700", "physrnn_wrapped_384", "physrnn_wrapped_2700", "physrnn_train_384", "physrnn_train_2700", "cur_gru128_2700", "cnn_384", "cnn_train_384",
       "cnn_train_512", "cnn_train_2700"]


def forward_leg(a, workload, rank, world, dist, steps, warmup, with_cpu):
    """One forward workload: value (whole job), HIP-event kernel times, roofline of the recurrent kernel."""
    import climsim_amd
    from synth import synth_inputs
    tag, B, flop_col, bytes_col = workload_spec(workload)
    consts, weights = load_model(tag)
    model = climsim_amd.NewModel_constraint(consts, weights, max_batch=B)
    stateful = model.stateful
    if a.halves >= 0:
        model.emulator.set_halves(bool(a.halves))   # default: automatic (on from 640 columns per GPU)
    # each rank owns its own shard of columns (different seed), resident in HBM
    xm, xs = synth_inputs(consts, B, 9000 + rank)
    rng = np.random.Generator(np.random.PCG64(100 + rank))
    hx, cx = rng.standard_normal((2, B, 128)).astype(np.float32)
    mem = np.zeros((B, 60, 16), np.float32) if stateful else None
    d_xm, d_xs = torch.from_numpy(xm).cuda(), torch.from_numpy(xs).cuda()
    d_hx, d_cx = torch.from_numpy(hx).cuda(), torch.from_numpy(cx).cuda()
    st = {"mem": torch.from_numpy(mem).cuda() if stateful else None}
    out = torch.empty(B, model.emulator.packed_width, device="cuda")

    def step():
        y = model.emulator.forward_packed(d_xm, d_xs, st["mem"], d_hx, d_cx, out=out)
        if stateful:   # caller-owned state, fed back exactly like the reference harness
            st["mem"] = y[:, 368:].reshape(B, 60, 16).contiguous()
        return y

    el, pre = _timed(step, steps, warmup, dist)
    value = world * B * steps / el

    # ---- per-kernel durations (HIP events on the launch stream), same step loop -------------------
    em = model.emulator
    em.set_profiling(True)
    em.reset_profile()
    for _ in range(100):
        step()
    prof, ncalls = em.get_profile()
    em.set_profiling(False)
    if rank != 0:
        return None
    rec_ms = 0.5 * (prof["rec_rnn1"] + prof["rec_rnn2"])
    rec_flop = B * 60 * 2.0 * 4 * 128 * 128          # algorithmic FLOP of one recurrent launch
    achieved = rec_flop / (rec_ms * 1e-3) / 1e12 if rec_ms > 0 else 0.0
    rec_name = rec_kernel_name(B)
    leg = {
        "value": value, "unit": "grid-columns/s", "steps": steps, "warmup": warmup, "preheat_steps": pre,
        "ms_per_step": 1e3 * el / steps,
        "config": {"workload": workload, "columns_per_gpu": B, "nlev": 60,
                   "wrapper": "stateless v4 (rnn/v4_rnn_wrapper_constrained.pt weights)" if not stateful
                   else "stateful v4 memory wrapper (rnn/v4_rnn-memory_wrapper_constrained_huber.pt weights), rnn1_mem fed back",
                   "parallelism": f"columns sharded x{world}, no collective"},
        "roofline": {"bound": "mfma",
                     "pipe": "v_mfma_f32_4x4x1 (dense fp32 matrix peak 157.3 TF)" if "rec4m" in rec_name
                     else "v_pk_fma_f32 (fp32 vector peak = f32 MFMA peak, 157.3 TF: the two share the FMA lanes)",
                     "kernel": rec_name + " (one launch per LSTM, 2 per step)",
                     "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_TFLOPS, "traffic": pmc_traffic(workload, rec_name.split("<")[0]),
                     "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/)",
                     "algorithmic_hbm_bytes_per_launch": B * 60 * (4 * 128 + 128) * 4.0,
                     "flop_per_launch": rec_flop, "avg_launch_ms": rec_ms},
        "whole_path": {"flop_per_column": flop_col, "achieved_tflops": value / world * flop_col / 1e12,
                       "frac_fp32_peak": value / world * flop_col / 1e12 / PEAK_FP32_TFLOPS,
                       "hbm_bytes_per_column": bytes_col,
                       "achieved_hbm_gbs": value / world * bytes_col / 1e9},
        "kernel_ms": prof, "kernel_profile_calls": ncalls,
    }
    if with_cpu:
        leg["cpu_baseline"] = cpu_baseline(tag, consts, weights, xm, xs, mem, hx, cx)
    model.emulator.close()
    return leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="default",
                    help="default | v4_stateless_<B> | v4_memory_<B> | train_tbptt3_<B> | " + " | ".join(AUX))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shard", action="store_true", help="default line without the shard_2700 / strong_scaled objects")
    ap.add_argument("--halves", type=int, default=-1, help="1/0: force the two-stream column-half path on/off (default: library default)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # one rank per GPU.  Rehearsal only (1-GPU box): CSA_BENCH_BACKEND=gloo lets several ranks share device 0 so that
    # the N>1 control flow (barriers, max-over-ranks, the gradient all-reduce) can be exercised without RCCL.
    backend = os.environ.get("CSA_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        sys.exit(f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    local_rank = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def finish():
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    if a.workload in AUX or a.workload.rsplit("_", 1)[0] in {w.rsplit("_", 1)[0] for w in AUX}:      # any column count: cur_lstm144_2700 ...
        aux_workload(a, rank, world, dist)
        return finish()
    with_cpu = not a.no_cpu_baseline and world == 1
    common = {"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    if "train_tbptt3_" in a.workload and not a.workload.startswith("cnn"):      # train_tbptt3_<B>, lstm144_train_tbptt3_<B>, gru128_train_tbptt3_<B>
        model = a.workload.split("train_tbptt3_")[0].rstrip("_")
        t = train_leg(a, rank, world, dist, a.steps, a.warmup, with_cpu and not model, B=int(a.workload.rsplit("_", 1)[1]),
                      tag="cur_" + (model or "lstm128"))
        if rank == 0:
            print(json.dumps({**t, **common}), flush=True)
        return finish()
    head = forward_leg(a, "v4_stateless_384" if a.workload == "default" else a.workload, rank, world, dist, a.steps, a.warmup, with_cpu)
    line = None
    if rank == 0:
        line = {"metric": "grid-columns/sec emulator fwd", **head, **common}
    if a.workload == "default":
        # the other half of BASELINE.json's metric and the north_star's target model, in the same line
        memleg = forward_leg(a, "v4_memory_384", rank, world, dist, a.steps, a.warmup, False)
        tsteps, twarm = max(3, min(a.steps, 30)), max(2, min(a.warmup, 5))
        train = train_leg(a, rank, world, dist, tsteps, twarm, with_cpu)
        if rank == 0:
            line["memory_wrapper"] = {"metric": "grid-columns/sec emulator fwd", **memleg}
            line["train"] = train
            if world > 1:
                line["cpu_baseline"] = carried_cpu_baseline("cpu_baseline")
                line["train"]["cpu_baseline"] = carried_cpu_baseline("train", "cpu_baseline")
        if not a.no_shard:
            # configs[3]/[4]: the share of ONE GPU when the 21,600-column high-resolution grid is split 8-way (2,700 columns per GPU,
            # whatever N is here) -- ms_per_step of `forward` is config 5's latency per global time step of that share
            sf = forward_leg(a, "v4_memory_2700", rank, world, dist, max(10, min(a.steps, 100)), max(2, min(a.warmup, 10)), False)
            st = train_leg(a, rank, world, dist, max(3, min(a.steps, 10)), 2, False, B=2700)
            # strong-scaled shapes of THIS run: the whole grid over the N GPUs that are here
            strong = {}
            shapes = [("hires_21600", 21600)] + ([("lowres_384", 384)] if world > 1 else [])
            for name, total in shapes:
                per = (total + world - 1) // world
                leg = forward_leg(a, f"v4_memory_{per}", rank, world, dist, max(5, min(a.steps, 50)), 3, False)
                if rank == 0:
                    strong[name] = {"global_columns": total, "columns_per_gpu": per, "scaling": "strong",
                                    "latency_ms_per_global_timestep": leg["ms_per_step"], "value": leg["value"], "unit": leg["unit"],
                                    "whole_path": leg["whole_path"], "kernel_ms": leg["kernel_ms"], "rec_kernel": leg["roofline"]["kernel"]}
            if rank == 0:
                line["shard_2700"] = {"forward": {"metric": "grid-columns/sec emulator fwd", **sf},
                                      "train": st}
                line["strong_scaled"] = strong
            # the physRNN side of the path: the deployed frozen export and the training step, 384 columns per GPU
            phys = {}
            for w in ("physrnn_wrapped_384", "physrnn_train_384"):
                aa = argparse.Namespace(**{**vars(a), "workload": w, "steps": max(10, min(a.steps, 100)), "warmup": max(3, min(a.warmup, 10))})
                o = aux_workload(aa, rank, world, dist, ret=True)
                if rank == 0:
                    phys[w] = {k: o[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config", "whole_path")}
            if rank == 0:
                line["physrnn"] = phys
            # opt-in projection GEMM (csa_set_gemm_split: fp32 operands split exactly into three bf16 values, six partial products on the
            # bf16 matrix pipe, fp32 accumulation -- DESIGN.md 4.10): the headline workload and the memory wrapper once more with it on.
            # NOT the headline value: the default library path (fp32 MFMA chain) is what `value` above measures.
            from climsim_amd import _lib
            _lib.lib().csa_set_gemm_split(1)
            try:
                sg = {w: forward_leg(a, w, rank, world, dist, a.steps, a.warmup, False) for w in ("v4_stateless_384", "v4_memory_384")}
            finally:
                _lib.lib().csa_set_gemm_split(0)
            if rank == 0:
                line["gemm_split_optin"] = {w: {k: o[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "kernel_ms", "whole_path")}
                                            for w, o in sg.items()}
    if rank == 0:
        print(json.dumps(line), flush=True)
    finish()


def carried_cpu_baseline(*path):
    """N > 1 lines: the CPU leg is timed on rank 0 of the N = 1 run only (it is a property of the host, not of N); carry the committed
    N = 1 measurement (profiles/r*_bench_default.json, newest round) with its provenance instead of re-timing it."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    if not fs:
        return None
    try:
        d = json.load(open(fs[-1]))
        for k in path:
            d = d[k]
        return {**d, "source": "N = 1 run, " + os.path.relpath(fs[-1], ROOT)}
    except Exception:
        return None


if __name__ == "__main__":
    main()
