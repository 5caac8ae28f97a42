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


def ensemble_window(model, x_lay, x_sfc, rnn_mem, ensemble_size, noise=None):
    """Ensemble forward over one window, as the reference's stochastic training / validation loop runs it
    (rnn/utils.py:1065-1075: every input is replicated `ensemble_size` times along the batch axis, member-major, and the
    members differ only through the noise the stochastic model draws; :1200-1215: the window's outputs are concatenated
    over time).  `model` is the model-level mirror (`wrappers.RNN_autoreg` built with add_stochastic_layer);
    x_lay (T, B, nlev, nx) and x_sfc (T, B, nx_sfc) normalised, rnn_mem (nlev, E*B, nh_mem) or None (zeros).
    noise: optional list over time of (hx0, cx0, eps) for the E*B replicated batch.
    Returns (preds_lay (T*E*B, nlev, ny), preds_sfc (T*E*B, ny_sfc), rnn_mem) with rows ordered (time, member, column),
    the layout `metrics.CRPS` / `compute_spread_skill_ratio` expect."""
    T, B = x_lay.shape[0], x_lay.shape[1]
    E = int(ensemble_size)
    c = model.emulator.cfg
    if rnn_mem is None:
        rnn_mem = torch.zeros(c.nlev, E * B, c.nh_mem, device=x_lay.device)
    lay, sfc = [], []
    for j in range(T):
        xl = x_lay[j].unsqueeze(0).expand(E, *x_lay[j].shape).reshape(E * B, *x_lay[j].shape[1:])   # repeat_interleave + flatten
        xs = x_sfc[j].unsqueeze(0).expand(E, *x_sfc[j].shape).reshape(E * B, *x_sfc[j].shape[1:])
        o, s, rnn_mem = model([xl, xs, rnn_mem], noise=None if noise is None else noise[j])
        lay.append(o)
        sfc.append(s)
    return torch.cat(lay), torch.cat(sfc), rnn_mem


def ensemble_scores(model, x_lay, x_sfc, targets_lay, targets_sfc, ensemble_size, rnn_mem=None, noise=None, beta=1, alpha=1.0):
    """One validation window of the stochastic model: ensemble forward, then the scores the reference logs for it
    (rnn/utils.py:1213-1215): CRPS (rnn/metrics.py:535-626) and the spread / skill pair (:509-533).
    targets_lay (T*B, nlev, ny), targets_sfc (T*B, ny_sfc) normalised.  Returns dict(crps, spread, rmse, rnn_mem)."""
    from . import metrics
    T = x_lay.shape[0]
    pl, ps, rnn_mem = ensemble_window(model, x_lay, x_sfc, rnn_mem, ensemble_size, noise=noise)
    crps = metrics.CRPS(targets_lay, targets_sfc, pl, ps, T, beta=beta, alpha=alpha)
    spread, rmse = metrics.compute_spread_skill_ratio(targets_lay, targets_sfc, pl, ps, T)
    return {"crps": crps, "spread": spread, "rmse": rmse, "rnn_mem": rnn_mem, "preds_lay": pl, "preds_sfc": ps}
