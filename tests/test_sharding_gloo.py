"""CPU, world_size 2 over gloo: the N>1 control flow (column shards, no data-path collective,
max-over-ranks timing, the one flat gradient all-reduce) -- and, through the oracle, that
sharded evaluation reproduces the unsharded result bit for bit (columns are independent)."""
import os
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
