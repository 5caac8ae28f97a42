"""SURVEY 8(e) on the HIP path with TWO REAL RANKS: `CNNTrainer.train_step(world_size=2)`, `MLPTrainer.train_step` and
`Trainer.ensemble_window_step(world_size=2, global_columns=...)` run as two processes that issue the one flat-gradient all-reduce
themselves (gloo on CUDA tensors, both ranks on device 0 of the one-GPU box; with two GPUs visible each rank takes its own and the
backend is RCCL) -- ragged column shards -- and the result equals the single-process step on the whole batch.  CPU twins of the same
properties, through the oracle, are in tests/test_sharding_gloo.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, load_npz_model
from climsim_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.set_num_threads(2)
    two = torch.cuda.device_count() >= world
    torch.cuda.set_device(rank if two else 0)
    if two:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)


def _cnn_case():
    from test_cnn_baseline import _arch
    depth, width, B = 3, 72, 11
    ws, bs = _arch(depth, width, seed=3)
    g = torch.Generator().manual_seed(17)
    x, yt = torch.randn(B, 60, 6, generator=g), torch.randn(B, 60, 10, generator=g)
    masks = (torch.rand(2 * depth, B, 60, width, generator=g) >= 0.175).to(torch.uint8)
    return depth, width, B, ws, bs, x, yt, masks


def _cnn_worker(rank, world, port, q):
    _init(rank, world, port)
    try:
        from climsim_amd.baselines import CNNTrainer
        depth, width, B, ws, bs, x, yt, masks = _cnn_case()
        tr = CNNTrainer([w.numpy() for w in ws], [b.numpy() for b in bs], depth=depth, width=width, dropout=0.175, max_batch=B)
        lo, hi = sharding.shard_bounds(B, world, rank)          # 6 + 5 columns
        m = masks[:, lo:hi].reshape(2 * depth, (hi - lo) * 60, width).contiguous().cuda()
        loss = tr.train_step(x[lo:hi].contiguous().cuda(), yt[lo:hi].contiguous().cuda(), masks=m, lr=1e-3, world_size=world, global_columns=B)
        torch.cuda.synchronize()
        if rank == 0:
            q.put((tr.grads.cpu(), float(loss), tr.flat_params().cpu()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(worker, n=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, n, port, q)) for r in range(n)]
    for p in procs:
        p.start()
    out = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


@pytest.mark.gpu
def test_cnn_train_step_two_ranks_equals_the_single_process_step():
    from climsim_amd.baselines import CNNTrainer
    g2, loss2, p2 = _run(_cnn_worker)
    depth, width, B, ws, bs, x, yt, masks = _cnn_case()
    tr = CNNTrainer([w.numpy() for w in ws], [b.numpy() for b in bs], depth=depth, width=width, dropout=0.175, max_batch=B)
    loss1 = tr.train_step(x.cuda(), yt.cuda(), masks=masks.reshape(2 * depth, B * 60, width).contiguous().cuda(), lr=1e-3)
    g1, p1 = tr.grads.cpu(), tr.flat_params().cpu()
    assert float((g2 - g1).abs().max()) <= 2e-6 * float(g1.abs().max())
    assert abs(loss2 - float(loss1)) <= 2e-6 * abs(float(loss1))
    # both took ONE Keras-Adam step from the same parameters: the first step is lr * g / (|g| + eps) per element, so it is compared
    # where the gradient is resolved (|g| well above the 2e-6 agreement of the two gradients)
    big = g1.abs() > 1e-3 * g1.abs().max()
    # (parameters are O(0.5): one fp32 ulp of the parameter itself, 6e-8, is the resolution of the comparison)
    assert float((p2 - p1)[big].abs().max()) <= 1e-5 * 1e-3 + 1.2e-7 and not torch.equal(p1, torch.zeros_like(p1))


def _ens_case():
    consts, weights, flags = load_npz_model("cur_stoch")
    grid = np.load(os.path.join(GOLDEN, "grid_consts.npz"))
    io = np.load(os.path.join(GOLDEN, "cur_stoch_train.npz"))
    return consts, weights, flags, grid, io


def _ens_step(tr, io, lo, hi, B, E, Tw, **kw):
    """Columns [lo, hi) of the golden ensemble window; member-major layouts (E, B, ...) are cut per member."""
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    n = hi - lo
    cols = lambda a, ax: np.take(a.reshape(a.shape[:ax] + (E, B) + a.shape[ax + 1:]), range(lo, hi), axis=ax + 1).reshape(
        a.shape[:ax] + (E * n,) + a.shape[ax + 1:])
    noise = [(d(cols(io[f"t{t}.hx0"], 0)), d(cols(io[f"t{t}.cx0"], 0)), d(cols(io[f"t{t}.eps"], 1))) for t in range(Tw)]
    tgt, tgt_sfc = io["tgt"], io["tgt_sfc"]
    return tr.ensemble_window_step([d(io[f"t{t}.x_main_n"][lo:hi]) for t in range(Tw)], [d(io[f"t{t}.x_sfc_n"][lo:hi]) for t in range(Tw)],
                                   [d(tgt[t * B + lo:t * B + hi]) for t in range(Tw)], [d(tgt_sfc[t * B + lo:t * B + hi]) for t in range(Tw)],
                                   d(cols(io["mem0"], 1)), E, noise=noise, optimise=False, **kw)


def _ens_worker(rank, world, port, q):
    _init(rank, world, port)
    try:
        from climsim_amd.train import Trainer
        consts, weights, flags, grid, io = _ens_case()
        B, E, Tw = int(io["B"]), int(io["E"]), int(io["T_w"])
        tr = Trainer(consts, weights, grid["hyai"], grid["hybi"], use_lstm=True, output_prune=bool(flags["output_prune"]), max_batch=B * E, max_window=Tw)
        lo, hi = sharding.shard_bounds(B, world, rank)
        sc, _, _ = _ens_step(tr, io, lo, hi, B, E, Tw, world_size=world, global_columns=B)
        torch.cuda.synchronize()
        if rank == 0:
            q.put((tr.grads.cpu(), sc))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_ensemble_window_step_two_ranks_equals_the_single_process_step_and_the_reference_gradient():
    from climsim_amd.train import Trainer
    g2, sc2 = _run(_ens_worker)
    consts, weights, flags, grid, io = _ens_case()
    B, E, Tw = int(io["B"]), int(io["E"]), int(io["T_w"])
    tr = Trainer(consts, weights, grid["hyai"], grid["hybi"], use_lstm=True, output_prune=bool(flags["output_prune"]), max_batch=B * E, max_window=Tw)
    sc1, _, _ = _ens_step(tr, io, 0, B, B, E, Tw)
    g1 = tr.grads.cpu()
    for name, (o, r, c) in tr.layout.items():
        a, b = g2[o:o + r * c], g1[o:o + r * c]
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-30, name
        ref = torch.from_numpy(io["dw." + name]).reshape(-1)
        assert float((a - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-30, name      # the reference's own autograd
    for k in ("loss", "skill", "spread"):
        assert abs(sc2[k] - sc1[k]) <= 2e-6 * abs(sc1[k]) + 1e-30, k


def _phys_case():
    from test_physrnn import _load
    from make_golden_physrnn import inputs
    from test_physrnn_train import _upstream
    g, P = _load()
    B = 13
    xm, xs, mem, xd = inputs(P, B, 77)
    hx2 = torch.randn(B, 128, generator=torch.Generator().manual_seed(5))
    return P, B, (xm, xs, mem, xd), hx2, _upstream(B, 3)


def _phys_step(P, B, lo, hi, inp, hx2, ups):
    from climsim_amd.physrnn import physical_RNN_autoreg, physical_RNN_trainer
    tr = physical_RNN_trainer(physical_RNN_autoreg(P, max_batch=B))
    c = lambda t: t[lo:hi].contiguous().cuda()
    tr.forward([c(t) for t in inp], hx2=c(hx2))
    tr.backward(*(c(u) for u in ups))
    return tr


def _phys_worker(rank, world, port, q):
    _init(rank, world, port)
    try:
        P, B, inp, hx2, ups = _phys_case()
        lo, hi = sharding.shard_bounds(B, world, rank)          # 7 + 6 columns
        tr = _phys_step(P, B, lo, hi, inp, hx2, ups)
        dist.all_reduce(tr.grads)                               # the ONE collective of the step: flat gradient, sum over the column shards
        tr.adam_step(1e-3)
        torch.cuda.synchronize()
        if rank == 0:
            q.put((tr.grads.cpu(), tr.params().cpu()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_physrnn_trainer_two_ranks_equal_the_single_process_step():
    """physRNN training shards by columns like the rest of the path: the loss is a sum over columns, so the flat gradients of ragged
    shards add up to the single-process gradient, and the Adam step taken from the all-reduced gradient is the same step."""
    g2, p2 = _run(_phys_worker)
    P, B, inp, hx2, ups = _phys_case()
    tr = _phys_step(P, B, 0, B, inp, hx2, ups)
    g1 = tr.grads.cpu()
    for name, off, r, c in tr.info:
        a, b = g2[off:off + r * c], g1[off:off + r * c]
        assert float((a - b).abs().max()) <= 5e-6 * float(b.abs().max()) + 1e-30, name
    tr.adam_step(1e-3)
    p1 = tr.params().cpu()
    big = g1.abs() > 1e-3 * g1.abs().max()
    assert float((p2 - p1)[big].abs().max()) <= 1e-5 * 1e-3 + 1.2e-7
