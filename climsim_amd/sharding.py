"""Column sharding for the multi-GPU path.

Grid columns are independent in the forward pass and in the carried memory, so N ranks split
them into contiguous blocks with NO data-path collective: rank r owns columns
[lo(r), hi(r)) of every global batch and keeps its own slice of rnn1_mem resident across
simulated time steps.  Only timing/diagnostics ever cross ranks (a scalar MAX / SUM), and
training adds ONE flat-buffer gradient all-reduce per optimiser step (see train.py).

These helpers contain no device code: they are shared by bench.py, the rollout driver and the
world_size-2 gloo tests that cover the N>1 control flow on CPU.
"""
import torch


def shard_bounds(n_columns, world_size, rank):
    """Contiguous, balanced split: the first (n % world) ranks get one extra column.
    384 -> 48 per GPU at 8 ranks; 21,600 -> 2,700 per GPU."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(int(n_columns), int(world_size))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_columns(t, world_size, rank, dim=0):
    lo, hi = shard_bounds(t.shape[dim], world_size, rank)
    return t.narrow(dim, lo, hi - lo)


def shard_loss_scale(local_columns, global_columns):
    """Factor that turns a batch-MEAN loss (or its gradient) over this rank's columns into this rank's share of the global
    batch mean, so that a SUM all-reduce over the ranks yields the single-GPU value (ragged shards included)."""
    return float(local_columns) / float(global_columns)


def max_over_ranks(seconds, device=None):
    """Whole-job wall time = the slowest rank's (bench.py contract)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_columns(local, world_size, sizes=None, dim=0):
    """Diagnostics only (e.g. writing one output file): all-gather the per-rank column blocks.
    Never on the data path of a step."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world_size == 1:
        return local
    if sizes is None:
        n = torch.tensor([local.shape[dim]], device=local.device)
        ns = [torch.zeros_like(n) for _ in range(world_size)]
        dist.all_gather(ns, n)
        sizes = [int(x.item()) for x in ns]
    mx = max(sizes)
    pad_shape = list(local.shape)
    pad_shape[dim] = mx
    buf = local.new_zeros(pad_shape)
    buf.narrow(dim, 0, local.shape[dim]).copy_(local)
    outs = [torch.empty_like(buf) for _ in range(world_size)]
    dist.all_gather(outs, buf)
    return torch.cat([o.narrow(dim, 0, s) for o, s in zip(outs, sizes)], dim=dim)


def allreduce_flat_(flat, world_size, average=True):
    """The single collective of a training step: SUM (then 1/world) over ONE contiguous fp32
    buffer holding every gradient (1.13 MB for the 283,629-parameter LSTM: latency-bound on
    xGMI, so one call on one buffer, never per parameter)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world_size == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat.mul_(1.0 / world_size)
    return flat
