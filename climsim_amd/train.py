"""Training step of the current-generation LSTM emulator (a13, a14, e).

Host-side mirror of the reference's TBPTT loop (rnn/utils.py:870-1600, condensed in SURVEY 3.2):

    for each window of T_w consecutive model time steps (B columns each):
        forward step by step, feeding rnn_mem back (graph kept)        utils.py:1098-1137
        loss = huber + w_hcon*energy + w_wcon*water over the window     utils.py:1203-1366
        zero_grad; backward; optimiser step; rnn_mem = rnn_mem.detach() utils.py:1363-1377,1580

Every FLOP (forward with saved activations, BPTT, loss and its gradient, Adam) runs in the HIP
kernels behind the C ABI; this class only sequences the calls and, with world_size > 1, issues the
ONE flat-buffer RCCL all-reduce per optimiser step (columns are sharded, weights replicated; the
batch-mean losses are divided by the GLOBAL column count so that summed shard gradients equal the
single-GPU gradient).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .emulator import CONST_KEYS, STATE_DICT_MAP, _check, _np32, _ptr
from .sharding import allreduce_flat_, shard_loss_scale


class Trainer:
    def __init__(self, consts, state_dict, hyai, hybi, *, use_lstm=True, output_prune=False, mp_mode=1, max_batch=384, max_window=3,
                 w_energy=6.0e-6, w_water=6.0e7, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self._h = None
        L = _lib.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("climsim_amd needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        host = {}
        params = _lib.CsaParams()
        for k in CONST_KEYS:
            host[k] = _np32(consts[k])
            setattr(params, k, host[k].ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        for sd, f in STATE_DICT_MAP.items():
            if sd in state_dict:
                host[f] = _np32(state_dict[sd])
                setattr(params, f, host[f].ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        cfg = _lib.CsaConfig()
        cfg.nlev, cfg.nx = host["xmean_lev"].shape
        cfg.nx_sfc = host["xmean_sca"].shape[0]
        cfg.ny = host["mlp_output_w"].shape[0]
        cfg.ny_sfc = host["mlp_surface_output_w"].shape[0]
        cfg.nh1 = host["rnn1_w_hh"].shape[1]
        # add_stochastic_layer (models.py:405-412) is recognised by its parameters: rnn0.* + rnn2.weight_encoder
        self.stochastic = "rnn2_weight_encoder" in host
        cfg.add_stochastic_layer = int(self.stochastic)
        cfg.nh2 = host["rnn2_weight_encoder"].shape[1] // 5 if self.stochastic else host["rnn2_w_hh"].shape[1]
        cfg.nh_mem = host["mlp_latent_w"].shape[0]
        cfg.use_lstm, cfg.legacy, cfg.mp_mode = int(use_lstm), 0, int(mp_mode)
        cfg.output_prune, cfg.scrub_inf = int(output_prune), 1
        self.cfg = cfg
        hyai, hybi = _np32(hyai), _np32(hybi)
        h = ctypes.c_void_p()
        fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        rc = L.csa_train_create(ctypes.byref(cfg), ctypes.byref(params), fp(hyai), fp(hybi), int(max_batch),
                                int(max_window), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_train_create failed ({rc}): {_lib.last_error()}")
        self._h = h
        self.max_batch, self.max_window = int(max_batch), int(max_window)
        self.w_energy, self.w_water = float(w_energy), float(w_water)
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), betas, float(eps), float(weight_decay)
        self.step_count = 0
        self.nparam = L.csa_train_num_params(h)
        self.layout = {}
        for i in range(L.csa_train_num_tensors(h)):
            name, off, r, c = ctypes.c_char_p(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            L.csa_train_param_info(h, i, ctypes.byref(name), ctypes.byref(off), ctypes.byref(r), ctypes.byref(c))
            self.layout[name.value.decode()] = (off.value, r.value, c.value)
        # ONE flat gradient buffer: what the kernels accumulate into and what the all-reduce moves
        self.grads = torch.zeros(self.nparam, device=self.device)
        self.scalars = torch.zeros(7, device=self.device)

    def close(self):
        if self._h is not None:
            _lib.lib().csa_train_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameter access (state_dict layout) --------------------------------------------------
    def _copy_state(self, which, buf=None):
        out = torch.empty(self.nparam, device=self.device) if buf is None else _check(buf, (self.nparam,), "state")
        self._rc(_lib.lib().csa_train_copy_state(self._h, which, 0 if buf is None else 1, _ptr(out), self._stream()),
                 "csa_train_copy_state")
        return out

    def flat_params(self):
        return self._copy_state(0)

    def state_dict(self):
        flat = self.flat_params()
        return {n: flat[o:o + r * c].reshape((r, c) if c > 1 or n.endswith("weight") else (r,)).clone()
                for n, (o, r, c) in self.layout.items()}

    def load_state_dict(self, sd):
        """Parameters from a state_dict with the reference's key names (resume / warm start, :761-794)."""
        flat = self.flat_params()
        for n, (o, r, c) in self.layout.items():
            flat[o:o + r * c] = torch.as_tensor(sd[n], dtype=torch.float32).to(self.device).reshape(-1)
        self._copy_state(0, flat)

    def checkpoint(self):
        """Everything a resume needs (cf. the torch.save dict of :1003-1009): parameters by name, Adam moments, step."""
        return {"model_state_dict": {k: v.cpu() for k, v in self.state_dict().items()},
                "adam_m": self._copy_state(1).cpu(), "adam_v": self._copy_state(2).cpu(), "step": self.step_count}

    def load_checkpoint(self, ck, only_load_model=False):
        self.load_state_dict(ck["model_state_dict"])
        if not only_load_model:
            self._copy_state(1, ck["adam_m"].to(self.device))
            self._copy_state(2, ck["adam_v"].to(self.device))
            self.step_count = int(ck["step"])

    def grad_dict(self):
        return {n: self.grads[o:o + r * c].reshape((r, c) if c > 1 or n.endswith("weight") else (r,))
                for n, (o, r, c) in self.layout.items()}

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _rc(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {_lib.last_error()}")

    # ---- per-stage timing (HIP events on the launch stream; bench.py's roofline accounting) -------------
    NSTAGE = 5

    def set_profiling(self, enable):
        self._rc(_lib.lib().csa_train_set_profiling(self._h, int(bool(enable))), "csa_train_set_profiling")

    def reset_profile(self):
        self._rc(_lib.lib().csa_train_reset_profile(self._h), "csa_train_reset_profile")

    def get_profile(self):
        """{stage: (average ms per bracketed launch group, number of groups)}"""
        L = _lib.lib()
        ms = (ctypes.c_double * self.NSTAGE)()
        n = (ctypes.c_long * self.NSTAGE)()
        self._rc(L.csa_train_get_profile(self._h, ms, n, self.NSTAGE), "csa_train_get_profile")
        return {L.csa_train_stage_name(i).decode(): (ms[i], n[i]) for i in range(self.NSTAGE)}

    # ---- pieces ------------------------------------------------------------------------------------
    def forward(self, slot, x_main_n, x_sfc_n, rnn_mem):
        c = self.cfg
        B = x_main_n.shape[0]
        x_main_n = _check(x_main_n, (B, c.nlev, c.nx), "x_main")
        x_sfc_n = _check(x_sfc_n, (B, c.nx_sfc), "x_sfc")
        rnn_mem = _check(rnn_mem, (c.nlev, B, c.nh_mem), "rnn_mem")
        out = torch.empty(B, c.nlev, c.ny, device=self.device)
        out_sfc = torch.empty(B, c.ny_sfc, device=self.device)
        mem_out = torch.empty(c.nlev, B, c.nh_mem, device=self.device)
        self._rc(_lib.lib().csa_train_forward(self._h, int(slot), B, _ptr(x_main_n), _ptr(x_sfc_n), _ptr(rnn_mem),
                                              _ptr(out), _ptr(out_sfc), _ptr(mem_out), self._stream()),
                 "csa_train_forward")
        return out, out_sfc, mem_out

    def forward_noise(self, slot, x_main_n, x_sfc_n, rnn_mem, hx0, cx0, eps):
        """Stochastic variant: forward of one window step with the reference's three randn draws given (hx0, cx0 (B,nh), eps
        (nlev,B,nh)); eps is kept referenced until the slot's backward."""
        c = self.cfg
        B = x_main_n.shape[0]
        x_main_n = _check(x_main_n, (B, c.nlev, c.nx), "x_main")
        x_sfc_n = _check(x_sfc_n, (B, c.nx_sfc), "x_sfc")
        rnn_mem = _check(rnn_mem, (c.nlev, B, c.nh_mem), "rnn_mem")
        hx0, cx0 = _check(hx0, (B, c.nh1), "hx0"), _check(cx0, (B, c.nh1), "cx0")
        eps = _check(eps, (c.nlev, B, c.nh2), "eps")
        out = torch.empty(B, c.nlev, c.ny, device=self.device)
        out_sfc = torch.empty(B, c.ny_sfc, device=self.device)
        mem_out = torch.empty(c.nlev, B, c.nh_mem, device=self.device)
        self._rc(_lib.lib().csa_train_forward_noise(self._h, int(slot), B, _ptr(x_main_n), _ptr(x_sfc_n), _ptr(rnn_mem), _ptr(hx0),
                                                    _ptr(cx0), _ptr(eps), _ptr(out), _ptr(out_sfc), _ptr(mem_out), self._stream()),
                 "csa_train_forward_noise")
        self._keep_eps = getattr(self, "_keep_eps", {})
        self._keep_eps[int(slot)] = eps
        return out, out_sfc, mem_out

    def backward(self, slot, d_out, d_out_sfc, d_mem_out=None, want_d_mem_in=True):
        c = self.cfg
        B = d_out.shape[0]
        d_out = _check(d_out, (B, c.nlev, c.ny), "d_out")
        d_out_sfc = _check(d_out_sfc, (B, c.ny_sfc), "d_out_sfc")
        if d_mem_out is not None:
            d_mem_out = _check(d_mem_out, (c.nlev, B, c.nh_mem), "d_mem_out")
        d_mem_in = torch.empty(c.nlev, B, c.nh_mem, device=self.device) if want_d_mem_in else None
        self._rc(_lib.lib().csa_train_backward(self._h, int(slot), B, _ptr(d_out), _ptr(d_out_sfc), _ptr(d_mem_out),
                                               _ptr(d_mem_in), _ptr(self.grads), self._stream()),
                 "csa_train_backward")
        return d_mem_in

    def set_deferred_wgrad(self, enable):
        """Postpone the W_ih / W_hh gradient GEMMs of backward() to flush_wgrad(): one contraction over all time steps
        of the window instead of one per step (window_step uses this)."""
        self._rc(_lib.lib().csa_train_set_deferred(self._h, int(bool(enable))), "csa_train_set_deferred")
        self._deferred = bool(enable)

    def flush_wgrad(self):
        self._rc(_lib.lib().csa_train_flush_wgrad(self._h, _ptr(self.grads), self._stream()), "csa_train_flush_wgrad")

    def loss(self, B, Tw, pred, pred_sfc, tgt, tgt_sfc, yto, yto_sfc, x_raw, x_sfc_n, with_grad=True):
        c = self.cfg
        N = B * Tw
        pred = _check(pred, (N, c.nlev, c.ny), "pred"); pred_sfc = _check(pred_sfc, (N, c.ny_sfc), "pred_sfc")
        tgt = _check(tgt, (N, c.nlev, c.ny), "tgt"); tgt_sfc = _check(tgt_sfc, (N, c.ny_sfc), "tgt_sfc")
        yto = _check(yto, (N, c.nlev, 6), "yto"); yto_sfc = _check(yto_sfc, (N, c.ny_sfc), "yto_sfc")
        x_raw = _check(x_raw, (N, c.nlev, c.nx), "x_raw"); x_sfc_n = _check(x_sfc_n, (N, c.nx_sfc), "x_sfc_n")
        d_pred = torch.empty_like(pred) if with_grad else None
        d_pred_sfc = torch.empty_like(pred_sfc) if with_grad else None
        self._rc(_lib.lib().csa_train_loss(self._h, B, Tw, self.w_energy, self.w_water, _ptr(pred), _ptr(pred_sfc),
                                           _ptr(tgt), _ptr(tgt_sfc), _ptr(yto), _ptr(yto_sfc), _ptr(x_raw),
                                           _ptr(x_sfc_n), _ptr(self.scalars), _ptr(d_pred), _ptr(d_pred_sfc),
                                           self._stream()), "csa_train_loss")
        return d_pred, d_pred_sfc

    def adam_step(self):
        self.step_count += 1
        self._rc(_lib.lib().csa_train_adam(self._h, _ptr(self.grads), self.lr, self.betas[0], self.betas[1], self.eps,
                                           self.weight_decay, self.step_count, self._stream()), "csa_train_adam")

    # ---- one TBPTT window = one optimiser step -----------------------------------------------------------
    def window_step(self, x_main_n, x_sfc_n, x_raw, tgt, tgt_sfc, yto, yto_sfc, rnn_mem, *, world_size=1,
                    global_columns=None, optimise=True):
        """x_* / targets: lists (or stacked tensors) over the T_w steps of the window, each (B, ...).
        Returns (scalars dict, new rnn_mem detached, d(rnn_mem at window start))."""
        Tw = len(x_main_n)
        B = x_main_n[0].shape[0]
        preds, preds_sfc, mem = [], [], rnn_mem
        for t in range(Tw):
            o, os_, mem = self.forward(t, x_main_n[t], x_sfc_n[t], mem)
            preds.append(o)
            preds_sfc.append(os_)
        cat = lambda xs: torch.cat(list(xs), 0).contiguous()
        d_pred, d_pred_sfc = self.loss(B, Tw, cat(preds), cat(preds_sfc), cat(tgt), cat(tgt_sfc), cat(yto),
                                       cat(yto_sfc), cat(x_raw), cat(x_sfc_n))
        if world_size > 1 and global_columns:
            # batch means were taken over the local shard: rescale so that SUM over ranks = global mean
            # (every loss term is a per-column quantity followed by a batch mean, rnn/metrics.py:142-315)
            scale = shard_loss_scale(B, global_columns)
            d_pred.mul_(scale)
            d_pred_sfc.mul_(scale)
            self.scalars.mul_(scale)
        self.grads.zero_()
        d_mem = None
        defer = Tw <= 8
        if defer != getattr(self, "_deferred", False):
            self.set_deferred_wgrad(defer)
        for t in reversed(range(Tw)):
            d_mem = self.backward(t, d_pred[t * B:(t + 1) * B], d_pred_sfc[t * B:(t + 1) * B], d_mem)
        if defer:
            self.flush_wgrad()
        if world_size > 1:
            allreduce_flat_(self.grads, world_size, average=not bool(global_columns))
            # logged scalars: the same weighting, one 28-byte all-reduce (train_mlp_h5loader.py:470-473 reduces its
            # logged loss likewise); without `global_columns` shards are equal and the plain average is the global mean
            allreduce_flat_(self.scalars, world_size, average=not bool(global_columns))
        if optimise:
            self.adam_step()
        names = ["loss", "huber", "mse", "mae", "energy", "water", "precip_sum_mse"]
        return dict(zip(names, self.scalars.tolist())), mem.detach(), d_mem

    # ---- ensemble / CRPS training of the stochastic variant (rnn/utils.py:1065-1075, 1213, 1363-1377) ------------------------
    def ensemble_window_step(self, x_main_n, x_sfc_n, tgt, tgt_sfc, rnn_mem, ensemble_size, *, noise=None, beta=1.0, alpha=1.0,
                             world_size=1, global_columns=None, optimise=True):
        """One optimiser step on the ensemble score: every step of the window runs E = ensemble_size members per column
        (inputs replicated member-major, one noise draw per member: `noise[t] = (hx0, cx0 (E*B,nh), eps (nlev,E*B,nh))`, drawn
        here when omitted), loss = CRPS(targets, ensemble outputs) over the window (rnn/metrics.py:535-626), its gradient comes
        from csa_crps_backward, BPTT runs through the slots with d(rnn_mem) chained.  rnn_mem: (nlev, E*B, nh_mem).
        Returns ({"loss": CRPS, "skill": ..., "spread": ...}, new rnn_mem (detached), d(rnn_mem at window start))."""
        if not self.stochastic:
            raise RuntimeError("ensemble_window_step needs a model with add_stochastic_layer")
        c, L = self.cfg, _lib.lib()
        Tw, B, E = len(x_main_n), x_main_n[0].shape[0], int(ensemble_size)
        BE = B * E
        rep = lambda t: torch.repeat_interleave(t.unsqueeze(0), E, dim=0).flatten(0, 1).contiguous()
        preds, preds_sfc, mem = [], [], rnn_mem
        for t in range(Tw):
            if noise is None:
                nz = (torch.randn(BE, c.nh1, device=self.device), torch.randn(BE, c.nh1, device=self.device),
                      torch.randn(c.nlev, BE, c.nh2, device=self.device))
            else:
                nz = noise[t]
            o, os_, mem = self.forward_noise(t, rep(x_main_n[t]), rep(x_sfc_n[t]), mem, *nz)
            preds.append(o)
            preds_sfc.append(os_)
        yp, yps = torch.cat(preds, 0).contiguous(), torch.cat(preds_sfc, 0).contiguous()
        y = _check(torch.cat(list(tgt), 0).contiguous(), (Tw * B, c.nlev, c.ny), "tgt")
        ys = _check(torch.cat(list(tgt_sfc), 0).contiguous(), (Tw * B, c.ny_sfc), "tgt_sfc")
        D1, D2 = c.nlev * c.ny, c.ny_sfc
        scratch, out3 = torch.empty(2 * Tw * B, device=self.device), torch.empty(3, device=self.device)
        self._rc(L.csa_crps(Tw, B, E, D1, D2, _ptr(y), _ptr(ys), _ptr(yp), _ptr(yps), float(beta), float(alpha), _ptr(scratch),
                            _ptr(out3), self._stream()), "csa_crps")
        share = shard_loss_scale(B, global_columns) if (world_size > 1 and global_columns) else 1.0
        d_p, d_s = torch.empty_like(yp), torch.empty_like(yps)
        self._rc(L.csa_crps_backward(Tw, B, E, D1, D2, _ptr(y), _ptr(ys), _ptr(yp), _ptr(yps), float(beta), float(alpha), float(share),
                                     _ptr(d_p), _ptr(d_s), self._stream()), "csa_crps_backward")
        self.grads.zero_()
        d_mem = None
        for t in reversed(range(Tw)):
            d_mem = self.backward(t, d_p[t * BE:(t + 1) * BE], d_s[t * BE:(t + 1) * BE], d_mem)
        sc = out3 * share
        if world_size > 1:
            allreduce_flat_(self.grads, world_size, average=not bool(global_columns))
            allreduce_flat_(sc, world_size, average=not bool(global_columns))
        if optimise:
            self.adam_step()
        v = sc.tolist()
        return {"loss": v[0], "skill": v[1], "spread": v[2]}, mem.detach(), d_mem
