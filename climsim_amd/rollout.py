"""Rollout harness: the reference's own evaluation loop (rnn/save_wrapper_mem.py:821-852) over a device-resident
time series, optionally column-sharded over the ranks of a torch.distributed job (every rank keeps
its slice of rnn1_mem; no collective on the data path)."""
import torch

from .sharding import shard_bounds


def rollout(model, xlev, xsfc, rnn1_mem=None, noise=None):
    """xlev (ntime, B, nlev, nx), xsfc (ntime, B, nx_sfc) on the GPU.  `model` is a NewModel_constraint (packed output,
    caller re-slices the state exactly as the reference harness does) or a model_wrapper (tuple output).
    Returns (outputs stacked over time, final rnn1_mem)."""
    ntime = xlev.shape[0]
    outs = []
    packed = hasattr(model, "stateful")
    if packed:
        B = xlev.shape[1]
        e = model.emulator.cfg
        nlev_mem, nmem = e.nlev, e.nh_mem
        if model.stateful and rnn1_mem is None:
            rnn1_mem = torch.zeros(B, nlev_mem, nmem, device=xlev.device)          # save_wrapper_mem.py:821
        for jj in range(ntime):                                                    # :827-852
            nz = None if noise is None else noise[jj]
            if model.stateful:
                out = model(xlev[jj], xsfc[jj], rnn1_mem, noise=nz)
                rnn1_mem = out[:, 368:368 + nlev_mem * nmem].reshape(B, nlev_mem, nmem)
            else:
                out = model(xlev[jj], xsfc[jj], noise=nz)
            outs.append(out[:, 0:368])
        return torch.stack(outs), rnn1_mem
    B = xlev.shape[1]
    c = model.emulator.cfg
    if rnn1_mem is None:
        rnn1_mem = torch.zeros(c.nlev, B, c.nh_mem, device=xlev.device)
    lev, sfc = [], []
    for jj in range(ntime):
        o, s, rnn1_mem = model(xlev[jj], xsfc[jj], rnn1_mem, noise=None if noise is None else noise[jj])
        lev.append(o)
        sfc.append(s)
    return (torch.stack(lev), torch.stack(sfc)), rnn1_mem


def sharded_rollout(model, xlev, xsfc, world_size, rank, **kw):
    """This rank's contiguous block of columns of a global (ntime, ncol, ...) series; state stays on the rank."""
    lo, hi = shard_bounds(xlev.shape[1], world_size, rank)
    return rollout(model, xlev[:, lo:hi].contiguous(), xsfc[:, lo:hi].contiguous(), **kw), (lo, hi)
