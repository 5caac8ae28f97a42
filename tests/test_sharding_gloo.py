"""CPU, world_size 2 over gloo: the N>1 control flow (column shards, no data-path collective,
max-over-ranks timing, the one flat gradient all-reduce) -- and, through the oracle, that
sharded evaluation reproduces the unsharded result bit for bit (columns are independent)."""
import os
import re
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, load_npz_model
from synth import synth_inputs
from climsim_amd import sharding


def test_shard_bounds_cover_and_balance():
    for n, w in ((384, 8), (21600, 8), (385, 2), (7, 8), (1, 1)):
        spans = [sharding.shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_bounds(384, 8, 3) == (144, 192)
    assert sharding.shard_bounds(21600, 8, 7) == (18900, 21600)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.pyoracle import OracleModel
        consts, weights, _ = load_npz_model("v4_memory")
        om = OracleModel(consts, weights, legacy=True)
        B = 11   # ragged split: 6 + 5
        xm, xs = synth_inputs(consts, B, 555)
        g = np.random.Generator(np.random.PCG64(9))
        mem = (0.3 * g.standard_normal((B, 60, 16))).astype(np.float32)
        hx, cx = g.standard_normal((2, B, 128)).astype(np.float32)
        lo, hi = sharding.shard_bounds(B, world, rank)
        y_local = om.wrapper_forward(xm[lo:hi], xs[lo:hi], mem[lo:hi], hx[lo:hi], cx[lo:hi])
        # diagnostics-only gather (never on the data path)
        y_all = sharding.gather_columns(torch.from_numpy(y_local), world).numpy()
        t = sharding.max_over_ranks(0.1 * (rank + 1))
        flat = torch.full((1000,), float(rank + 1))
        sharding.allreduce_flat_(flat, world)
        if rank == 0:
            y_ref = om.wrapper_forward(xm, xs, mem, hx, cx)
            q.put((np.array_equal(y_all, y_ref), t, float(flat[0]), y_all.shape))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_forward_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, t, g, shape = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same, "sharded evaluation must reproduce the unsharded rows exactly"
    assert shape == (11, 1328)
    assert abs(t - 0.2) < 1e-12          # MAX over ranks
    assert abs(g - 1.5) < 1e-6           # (1 + 2) / 2


def _grad_worker(rank, world, port, q):
    """Data-parallel gradient property of SURVEY 8(e): every loss term is a per-column quantity followed by a batch mean,
    so shard gradients of d(loss_local) * B_local / B_global, SUMMED by the one flat all-reduce, equal the unsharded
    gradient.  Checked with the autograd oracle on a ragged 5 + 3 split of the golden TBPTT window."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import torch_ref
        from test_train_golden import _golden, _window_inputs
        consts, weights, flags, io, grid = _golden("cur_lstm128")
        ref = torch_ref.EmulatorRef(consts, weights, legacy=False, use_lstm=True, output_prune=bool(flags["output_prune"]), scrub_inf=True)
        B, Tw, xr, xs, xn, xsn, tgt, tgt_sfc, yto, yto_sfc = _window_inputs(ref, io)
        lo, hi = sharding.shard_bounds(B, world, rank)
        rows = lambda t: torch.cat([t[k * B + lo:k * B + hi] for k in range(Tw)], 0)     # (Tw*B, ...) window tensors
        mem = torch.from_numpy(io["grad.mem0"])[:, lo:hi].contiguous()
        outs, outs_sfc = [], []
        for t in range(Tw):
            o, os_, mem = ref.model_forward(xn[t][lo:hi], xsn[t][lo:hi], mem)
            outs.append(o)
            outs_sfc.append(os_)
        loss, sc = torch_ref.window_loss(ref, torch.cat(outs, 0), torch.cat(outs_sfc, 0), rows(tgt), rows(tgt_sfc), rows(yto),
                                         rows(yto_sfc), torch.cat([x[lo:hi] for x in xr], 0), torch.cat([x[lo:hi] for x in xsn], 0),
                                         grid["hyai"], grid["hybi"], Tw)
        (loss * sharding.shard_loss_scale(hi - lo, B)).backward()
        names = [n for n, _ in ref.named_parameters()]
        flat = torch.cat([p.grad.reshape(-1) for _, p in ref.named_parameters()])
        scal = torch.tensor([float(sc["loss"])]) * sharding.shard_loss_scale(hi - lo, B)
        sharding.allreduce_flat_(flat, world, average=False)        # THE collective of a training step
        sharding.allreduce_flat_(scal, world, average=False)
        if rank == 0:
            gold = torch.cat([torch.from_numpy(io["grad.dw." + n]).reshape(-1) for n in names])
            worst, off = 0.0, 0
            for n, p in ref.named_parameters():
                k = p.numel()
                worst = max(worst, float((flat[off:off + k] - gold[off:off + k]).abs().max() / gold[off:off + k].abs().max()))
                off += k
            q.put((worst, float(scal[0]), float(io["grad.loss.loss"])))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_gradients_sum_to_the_unsharded_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    worst, loss, gold_loss = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert worst <= 2e-5, worst                       # summation order of the batch mean only (measured 2e-6)
    assert abs(loss - gold_loss) <= 2e-6 * abs(gold_loss)


def test_bench_spawns_its_own_ranks_before_touching_the_gpu():
    """`python bench.py --gpus 2` with no launcher: the parent starts two ranks under torch.distributed.run.  On this
    GPU-less container each rank stops at its device check, which proves the children ran as ranks 0 and 1 of 2."""
    import subprocess
    import sys
    root = os.path.dirname(GOLDEN.rstrip("/")).rsplit("/tests", 1)[0]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout
    else:
        # each rank stops at its own device check; the elastic agent may tear the other one down before it has printed, so
        # either rank's message -- under a 2-rank launch (WORLD_SIZE 2 is what makes bench.py reach that check) -- is the proof
        assert r.returncode != 0
        assert re.search(r"LOCAL_RANK [01] but only", r.stderr) or '"n_gpus": 2' in r.stdout, r.stderr[-2000:]
        assert "nproc" in r.stderr or "local_rank" in r.stderr or "torch.distributed" in r.stderr or "elastic" in r.stderr, r.stderr[-2000:]


def _cnn_ens_worker(rank, world, port, q):
    """CPU twins of tests/test_sharding_gpu.py through the oracle: (i) the Keras CNN's `mae_adjusted` and (ii) the ensemble score
    CRPS are per-column quantities followed by a batch mean, so ragged shard gradients scaled by B_local / B_global and SUMMED by the
    one flat all-reduce equal the unsharded gradient (CNNTrainer.train_step(world_size, global_columns),
    Trainer.ensemble_window_step(world_size, global_columns))."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import torch_ref
        from test_cnn_baseline import _arch
        from test_train_bigbatch import _torch_crps
        res = {}
        # (i) CNN
        depth, width, B = 2, 24, 7
        ws, bs = _arch(depth, width, seed=3)
        params = [t.clone().requires_grad_(True) for t in ws + bs]
        g = torch.Generator().manual_seed(17)
        x, yt = torch.randn(B, 60, 6, generator=g), torch.randn(B, 60, 10, generator=g)
        lo, hi = sharding.shard_bounds(B, world, rank)

        def cnn_loss(a, b):
            y = torch_ref.cnn_ref(x[a:b], params[:len(ws)], params[len(ws):], depth=depth)
            return torch_ref.mae_adjusted(yt[a:b], y)
        (cnn_loss(lo, hi) * sharding.shard_loss_scale(hi - lo, B)).backward()
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        sharding.allreduce_flat_(flat, world, average=False)
        if rank == 0:
            for p in params:
                p.grad = None
            cnn_loss(0, B).backward()
            full = torch.cat([p.grad.reshape(-1) for p in params])
            res["cnn"] = float((flat - full).abs().max() / full.abs().max())
        # (ii) CRPS over an ensemble window
        T, E, Bc = 2, 3, 5
        y, ys = torch.randn(T * Bc, 60, 5, generator=g), torch.randn(T * Bc, 8, generator=g)
        yp = torch.randn(T * E * Bc, 60, 5, generator=g).requires_grad_(True)
        yps = torch.randn(T * E * Bc, 8, generator=g).requires_grad_(True)
        lo, hi = sharding.shard_bounds(Bc, world, rank)
        n = hi - lo
        cut = lambda t, lead: t.reshape(lead + (Bc,) + t.shape[1:])[..., lo:hi, :, :].reshape((-1,) + t.shape[1:]) if t.dim() == 3 else \
            t.reshape(lead + (Bc,) + t.shape[1:])[..., lo:hi, :].reshape((-1,) + t.shape[1:])
        part = _torch_crps(cut(y, (T,)), cut(ys, (T,)), cut(yp, (T, E)), cut(yps, (T, E)), T) * sharding.shard_loss_scale(n, Bc)
        part.backward()
        flat = torch.cat([yp.grad.reshape(-1), yps.grad.reshape(-1)])
        val = part.detach().reshape(1).clone()
        sharding.allreduce_flat_(flat, world, average=False)
        sharding.allreduce_flat_(val, world, average=False)
        if rank == 0:
            yp.grad = None; yps.grad = None
            full_v = _torch_crps(y, ys, yp, yps, T)
            full_v.backward()
            full = torch.cat([yp.grad.reshape(-1), yps.grad.reshape(-1)])
            res["crps_grad"] = float((flat - full).abs().max() / full.abs().max())
            res["crps_val"] = abs(float(val) - float(full_v)) / abs(float(full_v))
            q.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_cnn_and_crps_shard_gradients_sum_to_the_unsharded_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cnn_ens_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["cnn"] <= 1e-5 and res["crps_grad"] <= 1e-5 and res["crps_val"] <= 1e-6, res


def _phys_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_physrnn import _load
        from test_physrnn_train import _autograd, _upstream
        from make_golden_physrnn import inputs
        g, P = _load()
        B = 9                                                   # ragged split: 5 + 4
        xm, xs, mem, xd = inputs(P, B, 21)
        hx2 = torch.randn(B, 128, generator=torch.Generator().manual_seed(4))
        ups = _upstream(B, 6)
        lo, hi = sharding.shard_bounds(B, world, rank)
        _, gl = _autograd(P, xm[lo:hi], xs[lo:hi], mem[lo:hi], xd[lo:hi], hx2[lo:hi], [u[lo:hi] for u in ups], torch.float64)
        names = sorted(k for k in gl if k != "rnn_mem")
        flat = torch.cat([gl[k].reshape(-1) for k in names])
        sharding.allreduce_flat_(flat, world, average=False)    # the ONE collective of a physRNN training step (sum: the loss is a sum over columns)
        if rank == 0:
            _, gw = _autograd(P, xm, xs, mem, xd, hx2, ups, torch.float64)
            ref = torch.cat([gw[k].reshape(-1) for k in names])
            q.put((float((flat - ref).abs().max()), float(ref.abs().max())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_physrnn_shard_gradients_sum_to_the_unsharded_gradient():
    """The physRNN training loss is a sum over columns: the flat gradients of two ragged column shards, all-reduced, are the gradient
    of the whole batch (CPU twin, through the restatement, of tests/test_sharding_gpu.py::test_physrnn_trainer_two_ranks_...)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_phys_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, scale = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert err <= 1e-10 * scale
