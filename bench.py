#!/usr/bin/env python3
"""bench.py -- grid-columns/s of the emulator forward on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): the v4 stateless wrapper (weights of
rnn/v4_rnn_wrapper_constrained.pt, committed as data under tests/golden/), one 384-column batch
of synthetic raw inputs per GPU per step, inputs resident in HBM.  Columns are independent, so
ranks shard them with NO data-path collective (weak scaling: 384 columns per GPU).

One JSON line on rank 0.  `roofline` is for the dominant kernel (the register-stationary
recurrent kernel, two launches per step), timed with HIP events on the launch stream;
`cpu_baseline` is the oracle's torch restatement (nn.LSTM on the host cores) on the same batch.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_TFLOPS = 157.3   # MI355X dense fp32 (vector == f32 MFMA), MI355X_MICROARCH.md
WORKLOADS = {
    # name: (model npz, columns per GPU, FLOP per column, HBM bytes per column)
    "v4_stateless_384": ("v4_stateless", 384, 31.79e6, 5148),
    "v4_memory_384": ("v4_memory", 384, 32.95e6, 12828),
    "v4_memory_2700": ("v4_memory", 2700, 32.95e6, 12828),
    "v4_memory_48": ("v4_memory", 48, 32.95e6, 12828),      # the low-res grid strong-scaled over 8 GPUs: latency case
}


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


def train_workload(a, rank, local_rank, world, dist):
    """Optional workload (BASELINE.json configs[2]): one TBPTT optimiser step of the current-generation
    LSTM (window T_w = 3, 384 columns per GPU): 3 forwards with saved activations, loss, 3 backwards,
    ONE flat-buffer RCCL all-reduce, Adam.  Unit: column-timesteps/s."""
    from climsim_amd.train import Trainer
    from oracle import torch_ref          # cpu_baseline leg only (rank 0, N=1)
    from synth import synth_inputs
    consts, weights = load_model("cur_lstm128")
    grid = np.load(os.path.join(ROOT, "tests", "golden", "grid_consts.npz"))
    B, Tw = 384, 3
    tr = Trainer(consts, weights, grid["hyai"], grid["hybi"], output_prune=True, max_batch=B, max_window=Tw)
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
    mem = torch.zeros(60, B, 16, device="cuda")

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        sc, mem, _ = tr.window_step(*args, mem, world_size=world, global_columns=B * world)
    fence()
    gc.collect(); gc.disable()     # a generation-2 collection over the interpreter's heap costs tens of ms: keep it out of the timed steps
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sc, mem, _ = tr.window_step(*args, mem, world_size=world, global_columns=B * world)
    fence()
    gc.enable()
    from climsim_amd.sharding import max_over_ranks
    el = max_over_ranks(time.perf_counter() - t0, device="cuda")
    if rank == 0:
        print(json.dumps({
            "metric": "train-step column-timesteps/sec (TBPTT window 3)", "value": world * B * Tw * a.steps / el,
            "unit": "column-timesteps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": a.workload, "columns_per_gpu": B, "window": Tw, "model": "RNN_autoreg LSTM 128/128, nh_mem 16",
                       "parallelism": f"columns sharded x{world}, one flat-gradient all-reduce per step"},
            "loss": sc["loss"]}), flush=True)


def aux_workload(a, rank, world, dist):
    """Secondary workloads (parity-test configurations, reported for DESIGN.md's tables; not the headline):
    cur_lstm144_384   current-generation tuple wrapper at the reference's default width nh = 144
    mlp_384 / cnn_384 offline Keras baselines (random weights of the reference architectures), forward."""
    import climsim_amd
    from climsim_amd.baselines import MLPBaseline, CNNBaseline
    from synth import synth_inputs
    B = 384
    g = torch.Generator().manual_seed(100 + rank)
    if a.workload in ("cur_lstm144_384", "cur_gru128_384", "cur_lstm128_384"):
        tag = a.workload[:-4]
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
    elif a.workload == "physrnn_384":
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
    elif a.workload == "cnn_train_384":
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

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    fence()
    gc.collect(); gc.disable()     # see train_workload: no interpreter garbage collection inside the timed steps
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    gc.enable()
    from climsim_amd.sharding import max_over_ranks
    el = max_over_ranks(time.perf_counter() - t0, device="cuda")
    if rank == 0:
        tf = B * flop_col * a.steps / el / 1e12
        print(json.dumps({
            "metric": "train-step columns/sec" if "train" in a.workload else "grid-columns/sec emulator fwd",
            "value": world * B * a.steps / el, "unit": "grid-columns/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": a.workload, "columns_per_gpu": B, "what": what,
                       "parallelism": f"columns sharded x{world}, no collective"},
            "whole_path": {"flop_per_column": flop_col, "achieved_tflops": tf, "frac_fp32_peak": tf / PEAK_FP32_TFLOPS}}),
            flush=True)


AUX = ["cur_lstm144_384", "cur_lstm128_384", "cur_gru128_384", "mlp_384", "online_mlp_384", "physrnn_384", "cnn_384", "cnn_train_384"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="v4_stateless_384", choices=sorted(WORKLOADS) + ["train_tbptt3_384"] + AUX)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--halves", type=int, default=-1, help="1/0: force the two-stream column-half path on/off (default: library default)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        sys.exit(f"--gpus {a.gpus} needs torch.distributed.run with WORLD_SIZE={a.gpus} (got {world})")
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

    import climsim_amd
    from synth import synth_inputs

    if a.workload in AUX:
        aux_workload(a, rank, world, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.workload == "train_tbptt3_384":
        train_workload(a, rank, local_rank, world, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    tag, B, flop_col, bytes_col = WORKLOADS[a.workload]
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
    d_mem = torch.from_numpy(mem).cuda() if stateful else None
    out = torch.empty(B, model.emulator.packed_width, device="cuda")

    def step():
        nonlocal d_mem
        y = model.emulator.forward_packed(d_xm, d_xs, d_mem, d_hx, d_cx, out=out)
        if stateful:   # caller-owned state, fed back exactly like the reference harness
            d_mem = y[:, 368:].reshape(B, 60, 16).contiguous()
        return y

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    gc.collect(); gc.disable()     # interpreter garbage collection (tens of ms for a full pass) stays outside the timed steps
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    el = time.perf_counter() - t0
    gc.enable()
    from climsim_amd.sharding import max_over_ranks
    el = max_over_ranks(el, device="cuda")     # whole-job time = the slowest rank's
    value = world * B * a.steps / el

    # ---- per-kernel durations (HIP events on the launch stream), same step loop -------------------
    em = model.emulator
    em.set_profiling(True)
    em.reset_profile()
    for _ in range(min(a.steps, 100)):
        step()
    prof, ncalls = em.get_profile()
    em.set_profiling(False)
    rec_ms = 0.5 * (prof["rec_rnn1"] + prof["rec_rnn2"])
    rec_flop = B * 60 * 2.0 * 4 * 128 * 128          # algorithmic FLOP of one recurrent launch
    achieved = rec_flop / (rec_ms * 1e-3) / 1e12 if rec_ms > 0 else 0.0

    if rank == 0:
        line = {
            "metric": "grid-columns/sec emulator fwd",
            "value": value, "unit": "grid-columns/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": a.workload, "columns_per_gpu": B, "nlev": 60,
                       "wrapper": "stateless v4 (rnn/v4_rnn_wrapper_constrained.pt weights)" if not stateful
                       else "stateful v4 memory wrapper", "parallelism": f"columns sharded x{world}, no collective"},
            "roofline": {"bound": "mfma", "kernel": ("lstm_rec1_kernel<128>" if B <= 256 else "lstm_rec2_kernel<128,false,0>") + " (one launch per LSTM, 2 per step)",
                         "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": pmc_traffic(a.workload),
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/)",
                         "algorithmic_hbm_bytes_per_launch": B * 60 * (4 * 128 + 128) * 4.0,
                         "flop_per_launch": rec_flop, "avg_launch_ms": rec_ms,
                         "note": "fp32 packed-FMA vector pipe; same 157.3 TF peak as f32 MFMA"},
            "whole_path": {"flop_per_column": flop_col, "achieved_tflops": value / world * flop_col / 1e12,
                           "frac_fp32_peak": value / world * flop_col / 1e12 / PEAK_FP32_TFLOPS,
                           "hbm_bytes_per_column": bytes_col,
                           "achieved_hbm_gbs": value / world * bytes_col / 1e9},
            "kernel_ms": prof, "kernel_profile_calls": ncalls,
        }
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(tag, consts, weights, xm, xs, mem, hx, cx)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
