"""Host mirror of the reference's physRNN "Hidden" model (rnn/models/models_phys.py::physical_RNN_autoreg, forward
:1586-1823) in the geometry of the shipped `rnn/saved_models/physRNN-Hidden_*_script_cpu.pt` artefacts.  Same call
convention as the TorchScript module: `forward([x_main_norm, x_sfc_norm, rnn_mem, x_denorm]) -> (out, out_sfc, rnn_mem)`;
the reference draws rnn2's initial state with torch.randn inside forward -- pass `hx2=` to make a call reproducible.
All arithmetic runs in the HIP library (`csa_phys_*`, csrc/phys.hip); there is no CPU fallback."""
import ctypes

import numpy as np
import torch

from . import _lib
from .emulator import _check, _ptr

_HEADS = ["mlp_qv_crm", "mlp_qn_crm", "mlp_t_crm", "mlp_subgrid_area_frac", "mlp_massflux", "mlp_eddy_diff", "mlp_qice_crm",
          "mlp_sed_qn_crm", "mlp_evap_prec_crm", "mlp_evap_cond_vapor_crm", "mlp_mp_aa_crm"]
_ORDER = (["hyam", "hybm", "hyai", "hybi", "yscale_lev", "yscale_sca", "xdiv_sca", "xmean_sca",
           "mlp_initial.weight", "mlp_initial.bias", "mlp_surface1.weight", "mlp_surface1.bias"]
          + [f"rnn{r}.{n}_l0" for r in (1, 2) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
          + [f"{m}.{n}" for m in ("mlp_latent", "mlp_output", "mlp_surface_output_rad", "mlp_output_rad", "mlp_precip_release")
             for n in ("weight", "bias")]
          + [f"{m}.{n}" for m in _HEADS for n in ("weight", "bias")])


# the radiation graphs (use_physrad, e.g. ..._num4050_BEST_script_cpu.pt): no radiative Linear heads, and the weights of the
# serialised radiative_transfer after the decoder heads (order of include/climsim_amd.h::csa_phys_rad_create)
_ORDER_RAD = ([k for k in _ORDER if not k.startswith(("mlp_surface_output_rad", "mlp_output_rad"))]
              + ["lbd_qn", "yscale_sca_rad", "sw_solar_weights", "gas_optics_model_lw.xmin", "gas_optics_model_lw.xmax",
                 "gas_optics_model_lw.ymean", "gas_optics_model_lw.ystd"]
              + [f"{m}.{n}" for m in ("gas_optics_model_lw.mlp1", "gas_optics_model_lw.mlp2", "gas_optics_model_lw.mlp3",
                                      "gas_optics_lw_reduce1", "gas_optics_lw_reduce2", "mlp_sw_optprops1", "mlp_sw_optprops2")
                 for n in ("weight", "bias")])


# Slingo (1989) liquid and Ebert-Curry ice SW cloud-optics coefficients in their four bands (rnn/models/physics_rad_e3sm.py:108-113,
# :267-272), and the last RRTMGP g-point (of 112) of bands 4, 3, 2 as the shipped exports serialise the band -> g-point spread
_SLINGO = [[2.817e-02, 2.682e-02, 2.264e-02, 1.281e-02], [1.305, 1.346, 1.454, 1.641], [-5.62e-08, -6.94e-06, 4.64e-04, 0.201],
           [1.63e-07, 2.35e-05, 1.24e-03, 7.56e-03], [0.829, 0.794, 0.754, 0.826], [2.482e-03, 4.226e-03, 6.560e-03, 4.353e-03]]
_EBERT_CURRY = [[3.448e-03] * 4, [2.431] * 4, [1.00e-05, 1.10e-04, 1.861e-02, 0.46658], [0.0, 1.405e-05, 8.328e-04, 2.05e-05],
                [0.7661, 0.7730, 0.794, 0.9595], [5.851e-04, 5.665e-04, 7.267e-04, 1.076e-04]]
_SW_BAND_LIMITS = (37, 71, 80)


def _sw_gas_block(state_dict, ng, band_limits=_SW_BAND_LIMITS):
    """The CSA_PHYS_SW_GAS weight block of include/climsim_amd.h (layout: csrc/phys.h SWG_*)."""
    f = lambda k: np.asarray(state_dict[k].detach().cpu() if isinstance(state_dict[k], torch.Tensor) else state_dict[k], np.float32)
    if ng != 16:
        raise RuntimeError("physRNN SW gas optics: built for 16 g-points")
    pad8 = lambda v: np.concatenate([v, np.ones(8 - v.shape[0], np.float32)])
    parts = [pad8(f("gas_optics_model_sw1.xmin")) * np.r_[np.ones(7), 0].astype(np.float32), pad8(f("gas_optics_model_sw1.xdiv"))]
    padk = lambda a: np.concatenate([a, np.zeros((128 - a.shape[0],) + a.shape[1:], np.float32)])      # the 112-wide axis -> 128
    padc = lambda a, ld: np.concatenate([a, np.zeros((a.shape[0], ld - a.shape[1]), np.float32)], 1).ravel()   # row stride K + 4 (LDS banks)
    for m in ("gas_optics_model_sw1", "gas_optics_model_sw2"):
        w1 = f(m + ".mlp1.weight")
        if w1.shape != (32, 7) or f(m + ".mlp2.weight").shape != (32, 32) or f(m + ".mlp3.weight").shape != (112, 32):
            raise RuntimeError("physRNN SW gas optics: built for the shipped 7 -> 32 -> 32 -> 112 models")
        parts += [padc(w1, 12), f(m + ".mlp1.bias"), padc(f(m + ".mlp2.weight"), 36), f(m + ".mlp2.bias"),
                  padc(padk(f(m + ".mlp3.weight")), 36), padk(f(m + ".mlp3.bias")), padk(f(m + ".ystd")), padk(f(m + ".ymean"))]
    for r in ("gas_optics_sw_reduce1", "gas_optics_sw_reduce2"):
        parts += [padc(f(r + ".weight"), 132), f(r + ".bias")]
    b4, b3, b2 = (int(round(l / 112 * ng)) for l in band_limits)
    idx = [3] * b4 + [2] * (b3 - b4) + [1] * (b2 - b3) + [0] * (ng - b2)
    parts.append(np.asarray(_SLINGO + _EBERT_CURRY, np.float32)[:, idx].ravel())
    blk = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).ravel() for p in parts]), np.float32)
    assert blk.size == 17648, blk.size
    return blk


class physical_RNN_autoreg(torch.nn.Module):
    def __init__(self, state_dict, *, ilev_crm=10, mp_ncol=None, nh_mem0=15, max_batch=4096):
        super().__init__()
        self._h = None
        if not torch.cuda.is_available():
            raise RuntimeError("climsim_amd needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.use_physrad = "gas_optics_model_lw.mlp1.weight" in state_dict
        if mp_ncol is None:
            mp_ncol = int(state_dict["mlp_qv_crm.weight"].shape[0]) if "mlp_qv_crm.weight" in state_dict else 16
        if self.use_physrad:
            self._init_rad(state_dict, ilev_crm, mp_ncol, nh_mem0, max_batch)
            return
        arrs = []
        for k in _ORDER:
            if k not in state_dict:
                raise RuntimeError(f"physRNN state_dict lacks {k}")
            v = state_dict[k]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            arrs.append(np.ascontiguousarray(v, np.float32))
        sd = dict(zip(_ORDER, arrs))
        self.nh = sd["mlp_initial.weight"].shape[0]
        self.nx = sd["mlp_initial.weight"].shape[1] - 1
        self.nx_sfc = sd["mlp_surface1.weight"].shape[1]
        self.nlev, self.nlev_mem, self.nh_mem = 60, 60 - ilev_crm, nh_mem0 + 1
        FP = ctypes.POINTER(ctypes.c_float)
        warr = (FP * len(arrs))(*[a.ctypes.data_as(FP) for a in arrs])
        h = ctypes.c_void_p()
        rc = _lib.lib().csa_phys_create(self.nx, self.nx_sfc, self.nh, int(ilev_crm), int(mp_ncol), int(nh_mem0), warr,
                                        int(max_batch), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_phys_create failed ({rc}): {_lib.last_error()}")
        self._h, self.max_batch = h, max_batch

    def _init_rad(self, state_dict, ilev_crm, mp_ncol, nh_mem0, max_batch):
        arrs = []
        ng = int(state_dict["gas_optics_lw_reduce1.weight"].shape[0])
        self.stochastic = "rnn3.weight_ih" in state_dict
        # the physRNN_physRad-* generation: clear-sky region (one row fewer in the two condensate heads), no sub-grid temperature head,
        # rnn_mem level-major (50, B, 16)
        self.physrad = "mlp_t_crm.weight" not in state_dict and int(state_dict["mlp_qn_crm.weight"].shape[0]) == mp_ncol - 1
        flags = ((1 if mp_ncol != ng else 0) | (2 if "mlp_liq_frac_crm.weight" in state_dict else 0) | (4 if self.stochastic else 0)
                 | (8 if self.physrad else 0) | (16 if "gas_optics_model_lw.xdiv" in state_dict else 0)
                 | (32 if "cloud_optics_lw.weight" in state_dict else 0)
                 | (64 if "gas_optics_model_sw1.mlp1.weight" in state_dict else 0)
                 | (128 if "cloud_optics_sw.weight" in state_dict else 0))
        order = (_ORDER_RAD + (["mlp_liq_frac_crm.weight", "mlp_liq_frac_crm.bias"] if flags & 2 else [])
                 + (["rnn3.weight_ih", "rnn3.weight_zh", "rnn3.weight_encoder"] if flags & 4 else [])
                 + (["cloud_optics_lw.weight", "cloud_optics_lw.bias"] if flags & 32 else []))
        for k in order:
            if flags & 16 and k == "gas_optics_model_lw.xmax":       # later exports: the range itself travels in this slot
                k = "gas_optics_model_lw.xdiv"
            if (self.physrad and k.startswith("mlp_t_crm.")) or (flags & 64 and k.startswith("mlp_sw_optprops")):
                arrs.append(None)
                continue
            if k not in state_dict:
                raise RuntimeError(f"physRNN (radiation graph) state_dict lacks {k}")
            v = state_dict[k]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            arrs.append(np.ascontiguousarray(v, np.float32))
        if flags & 64:
            arrs.append(_sw_gas_block(state_dict, ng))
        if flags & 128:         # two Linear layers without an activation between them: composed here (float64, rounded once)
            f64 = lambda k: np.asarray(state_dict[k].detach().cpu() if isinstance(state_dict[k], torch.Tensor) else state_dict[k], np.float64)
            w1, b1, w2, b2 = (f64("cloud_optics_sw" + k) for k in (".weight", ".bias", "2.weight", "2.bias"))
            arrs += [np.ascontiguousarray(w2 @ w1, np.float32), np.ascontiguousarray(w2 @ b1 + b2, np.float32)]
        sd = dict(zip(order, arrs))
        self.nh = sd["mlp_initial.weight"].shape[0]
        self.nx = sd["mlp_initial.weight"].shape[1] + 2             # x_main columns: 3 of them bypass mlp_initial, which also sees pressure
        self.nx_sfc = sd["xmean_sca"].shape[0]                      # x_sfc columns (5 of them bypass mlp_surface1)
        self.nlev, self.nlev_mem, self.nh_mem = 60, 60 - ilev_crm, nh_mem0 + 1
        FP = ctypes.POINTER(ctypes.c_float)
        warr = (FP * len(arrs))(*[a.ctypes.data_as(FP) if a is not None else None for a in arrs])
        h = ctypes.c_void_p()
        rc = _lib.lib().csa_phys_rad_create(self.nx, self.nx_sfc, self.nh, int(ilev_crm), int(mp_ncol), int(nh_mem0), ng, flags, warr,
                                            int(max_batch), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_phys_rad_create failed ({rc}): {_lib.last_error()}")
        self._h, self.max_batch = h, max_batch

    def forward(self, inp_list, hx2=None, hx1=None, eps3=None, _srnn=None):
        """hx2 (B, nh): rnn2's initial state; add_stochastic_layer graphs also take hx1 (B, nh), rnn3's initial state, and eps3
        (50, B, nh), its noise.  Whatever is not passed is drawn here with torch.randn, as the reference does inside forward."""
        x_main, x_sfc, rnn_mem, x_denorm = inp_list[0], inp_list[1], inp_list[2], inp_list[3]
        B = x_main.shape[0]
        x_main = _check(x_main, (B, self.nlev, self.nx), "inputs_main")
        x_sfc = _check(x_sfc, (B, self.nx_sfc), "inputs_aux")
        mem_shape = (self.nlev_mem, B, self.nh_mem) if getattr(self, "physrad", False) else (B, self.nlev_mem, self.nh_mem)
        rnn_mem = _check(rnn_mem, mem_shape, "rnn_mem")
        x_denorm = _check(x_denorm, (B, self.nlev, x_denorm.shape[-1]), "inputs_denorm")
        hx2 = torch.randn(B, self.nh, device=self.device) if hx2 is None else _check(hx2, (B, self.nh), "hx2")
        out = torch.empty(B, self.nlev, 5, device=self.device)
        out_sfc = torch.empty(B, 8, device=self.device)
        mem_out = torch.empty(*mem_shape, device=self.device)
        if _srnn is not None:                 # test hook: rnn3's output supplied (teacher forcing), see csa_phys_debug_forward_srnn
            _srnn = _check(_srnn, (self.nlev_mem, B, self.nh), "srnn")
            rc = _lib.lib().csa_phys_debug_forward_srnn(self._h, B, _ptr(x_main), _ptr(x_sfc), _ptr(rnn_mem), _ptr(x_denorm),
                                                        int(x_denorm.shape[-1]), _ptr(hx2), _ptr(_srnn), _ptr(out), _ptr(out_sfc), _ptr(mem_out),
                                                        ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
            if rc != 0:
                raise RuntimeError(f"csa_phys_debug_forward_srnn failed ({rc}): {_lib.last_error()}")
            return out, out_sfc, mem_out
        if getattr(self, "stochastic", False):
            hx1 = torch.randn(B, self.nh, device=self.device) if hx1 is None else _check(hx1, (B, self.nh), "hx1")
            eps3 = torch.randn(self.nlev_mem, B, self.nh, device=self.device) if eps3 is None else _check(eps3, (self.nlev_mem, B, self.nh), "eps3")
            p1, p3 = _ptr(hx1), _ptr(eps3)
        else:
            p1 = p3 = None
        rc = _lib.lib().csa_phys_forward_noise(self._h, B, _ptr(x_main), _ptr(x_sfc), _ptr(rnn_mem), _ptr(x_denorm),
                                               int(x_denorm.shape[-1]), _ptr(hx2), p1, p3, _ptr(out), _ptr(out_sfc), _ptr(mem_out),
                                               ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"csa_phys_forward failed ({rc}): {_lib.last_error()}")
        return out, out_sfc, mem_out

    def postprocessing(self, out, out_sfc, x_denorm):
        """The module's exported method (rnn/models/models.py:273-339): (out (B,60,5), out_sfc (B,8)) normalised ->
        (out (B,60,6) physical with the cloud-water tendency split into liquid / ice, out_sfc / yscale_sca)."""
        B = out.shape[0]
        out, out_sfc = _check(out, (B, self.nlev, 5), "out"), _check(out_sfc, (B, 8), "out_sfc")
        x_denorm = _check(x_denorm, (B, self.nlev, x_denorm.shape[-1]), "x_denorm")
        out6, sfc = torch.empty(B, self.nlev, 6, device=self.device), torch.empty(B, 8, device=self.device)
        rc = _lib.lib().csa_phys_postprocess(self._h, B, _ptr(out), _ptr(out_sfc), _ptr(x_denorm), int(x_denorm.shape[-1]), _ptr(out6), _ptr(sfc),
                                             ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"csa_phys_postprocess failed ({rc}): {_lib.last_error()}")
        return out6, sfc

    def debug_rnn3(self, x, h0, eps):
        """Test hook: this graph's stochastic third RNN alone, x (T, B, nh), h0 (B, nh), eps (T, B, nh) -> (T, B, nh)."""
        T, B = x.shape[0], x.shape[1]
        x, h0, eps = _check(x, (T, B, self.nh), "x"), _check(h0, (B, self.nh), "h0"), _check(eps, (T, B, self.nh), "eps")
        out = torch.empty(T, B, self.nh, device=self.device)
        rc = _lib.lib().csa_phys_debug_rnn3(self._h, T, B, _ptr(x), _ptr(h0), _ptr(eps), _ptr(out),
                                            ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"csa_phys_debug_rnn3 failed ({rc}): {_lib.last_error()}")
        return out

    def tap(self, which, B):
        t = torch.empty(self.nlev_mem if self.use_physrad else 60, B, self.nh, device=self.device)
        rc = _lib.lib().csa_phys_tap(self._h, which, B, _ptr(t), ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"csa_phys_tap failed ({rc})")
        return t

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().csa_phys_destroy(self._h)
                self._h = None
        except Exception:
            pass


# ---- the frozen `*_wrapped` exports ---------------------------------------------------------------------------------------------------
class physical_RNN_trainer:
    """Training of the non-radiative physRNN graph on the device (C-ABI csa_phys_train_*; reference: the model the training script
    builds at rnn/train_rnn_rollout_torchscript_hydra.py:553-554 and differentiates with autograd).  The parameters live in ONE flat
    device vector in state_dict order; `forward` keeps the activations, `backward` takes the loss gradients w.r.t. the three outputs,
    accumulates into `self.grads` and returns the gradient w.r.t. the incoming memory (the link of a TBPTT window)."""

    def __init__(self, model, slots=1):
        if model.use_physrad:
            raise RuntimeError("physical_RNN_trainer: built for the non-radiative graph")
        self.model, self._h, self.device, self.slots = model, model._h, model.device, int(slots)
        L = _lib.lib()
        rc = L.csa_phys_train_enable(self._h, self.slots)
        if rc != 0:
            raise RuntimeError(f"csa_phys_train_enable failed ({rc}): {_lib.last_error()}")
        nt, nf = ctypes.c_int(), ctypes.c_int()
        L.csa_phys_train_num_params(self._h, ctypes.byref(nt), ctypes.byref(nf))
        self.nparam, self.info = nf.value, []
        for i in range(nt.value):
            name, off, rows, cols = ctypes.c_char_p(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            L.csa_phys_train_param_info(self._h, i, ctypes.byref(name), ctypes.byref(off), ctypes.byref(rows), ctypes.byref(cols))
            self.info.append((name.value.decode(), off.value, rows.value, cols.value))
        self.grads = torch.zeros(self.nparam, device=self.device)
        self._pending = {}

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def named(self, flat):
        """name -> view of a flat vector, shaped like the reference's state_dict entry"""
        return {n: (flat[o:o + r * c].view(r, c) if "weight" in n else flat[o:o + r * c]) for n, o, r, c in self.info}

    def params(self):
        flat = torch.empty(self.nparam, device=self.device)
        rc = _lib.lib().csa_phys_train_get_params(self._h, _ptr(flat), self._stream())
        if rc != 0:
            raise RuntimeError(f"csa_phys_train_get_params failed ({rc}): {_lib.last_error()}")
        return flat

    def set_params(self, flat):
        flat = _check(flat, (self.nparam,), "params")
        rc = _lib.lib().csa_phys_train_set_params(self._h, _ptr(flat), self._stream())
        if rc != 0:
            raise RuntimeError(f"csa_phys_train_set_params failed ({rc}): {_lib.last_error()}")

    def state_dict(self):
        return {k: v.clone() for k, v in self.named(self.params()).items()}

    def zero_grad(self):
        self.grads.zero_()

    def forward(self, inp_list, hx2=None, slot=0):
        m = self.model
        x_main, x_sfc, rnn_mem, x_denorm = inp_list[0], inp_list[1], inp_list[2], inp_list[3]
        B = x_main.shape[0]
        x_main = _check(x_main, (B, m.nlev, m.nx), "inputs_main")
        x_sfc = _check(x_sfc, (B, m.nx_sfc), "inputs_aux")
        rnn_mem = _check(rnn_mem, (B, m.nlev_mem, m.nh_mem), "rnn_mem")
        x_denorm = _check(x_denorm, (B, m.nlev, x_denorm.shape[-1]), "inputs_denorm")
        hx2 = torch.randn(B, m.nh, device=self.device) if hx2 is None else _check(hx2, (B, m.nh), "hx2")
        out, out_sfc = torch.empty(B, m.nlev, 5, device=self.device), torch.empty(B, 8, device=self.device)
        mem_out = torch.empty(B, m.nlev_mem, m.nh_mem, device=self.device)
        rc = _lib.lib().csa_phys_train_forward(self._h, int(slot), B, _ptr(x_main), _ptr(x_sfc), _ptr(rnn_mem), _ptr(x_denorm), int(x_denorm.shape[-1]),
                                               _ptr(hx2), _ptr(out), _ptr(out_sfc), _ptr(mem_out), self._stream())
        if rc != 0:
            raise RuntimeError(f"csa_phys_train_forward failed ({rc}): {_lib.last_error()}")
        self._pending[int(slot)] = (B, x_main, x_sfc, rnn_mem, x_denorm)
        return out, out_sfc, mem_out

    def backward(self, d_out, d_out_sfc, d_mem_out, slot=0):
        if int(slot) not in self._pending:
            raise RuntimeError("physical_RNN_trainer.backward: no pending forward in this slot")
        B, x_main, x_sfc, rnn_mem, x_denorm = self._pending.pop(int(slot))
        m = self.model
        d_out, d_out_sfc = _check(d_out, (B, m.nlev, 5), "d_out"), _check(d_out_sfc, (B, 8), "d_out_sfc")
        d_mem_out = _check(d_mem_out, (B, m.nlev_mem, m.nh_mem), "d_mem_out")
        d_mem_in = torch.empty(B, m.nlev_mem, m.nh_mem, device=self.device)
        rc = _lib.lib().csa_phys_train_backward(self._h, int(slot), B, _ptr(x_main), _ptr(x_sfc), _ptr(rnn_mem), _ptr(x_denorm), int(x_denorm.shape[-1]),
                                                _ptr(d_out), _ptr(d_out_sfc), _ptr(d_mem_out), _ptr(d_mem_in), _ptr(self.grads), self._stream())
        if rc != 0:
            raise RuntimeError(f"csa_phys_train_backward failed ({rc}): {_lib.last_error()}")
        return d_mem_in

    def window(self, steps, rnn_mem, loss_grad, hx2=None):
        """One truncated-BPTT window (the reference's loop, rnn/utils.py:1200-1377: the memory returned by step t is the input of
        step t+1, the loss is summed over the window, one backward).  steps: list of (x_main, x_sfc, x_denorm) per step (len <= slots);
        loss_grad(t, out, out_sfc) -> (d_out, d_out_sfc), the loss gradient of step t's outputs; hx2: list of rnn2 initial states or
        None (drawn).  Returns (outputs per step, final memory detached, d(loss)/d(incoming memory)); gradients accumulate in .grads."""
        if len(steps) > self.slots:
            raise RuntimeError("physical_RNN_trainer.window: more steps than slots")
        outs, mem = [], rnn_mem
        for t, (xm, xs, xd) in enumerate(steps):
            o, osfc, mem = self.forward([xm, xs, mem, xd], hx2=None if hx2 is None else hx2[t], slot=t)
            outs.append((o, osfc))
        d_mem = torch.zeros_like(mem)
        for t in reversed(range(len(steps))):
            d_o, d_sfc = loss_grad(t, *outs[t])
            d_mem = self.backward(d_o, d_sfc, d_mem, slot=t)
        return outs, mem, d_mem

    def loss(self, preds, preds_sfc, tgt, tgt_sfc, yto, yto_sfc, x_raw, x_sfc_n, Tw=1, w_energy=6.0e-6, w_water=6.0e7, grad=True, scalars=True):
        """The reference trainer's loss over a window of Tw steps stacked on the leading axis (rnn/utils.py:1203-1366) -> (dict of the
        seven scalars, d loss / d preds, d loss / d preds_sfc)."""
        N = preds.shape[0]
        B, m = N // Tw, self.model
        preds, preds_sfc = _check(preds, (N, m.nlev, 5), "preds"), _check(preds_sfc, (N, 8), "preds_sfc")
        tgt, tgt_sfc = _check(tgt, (N, m.nlev, 5), "tgt"), _check(tgt_sfc, (N, 8), "tgt_sfc")
        yto, yto_sfc = _check(yto, (N, m.nlev, 6), "yto"), _check(yto_sfc, (N, 8), "yto_sfc")
        x_raw = _check(x_raw, (N, m.nlev, x_raw.shape[-1]), "x_raw")
        x_sfc_n = _check(x_sfc_n, (N, m.nx_sfc), "x_sfc_n")
        sc = torch.empty(7, device=self.device)
        d_p, d_s = (torch.empty_like(preds), torch.empty_like(preds_sfc)) if grad else (None, None)
        rc = _lib.lib().csa_phys_train_loss(self._h, B, int(Tw), int(x_raw.shape[-1]), float(w_energy), float(w_water), _ptr(preds), _ptr(preds_sfc),
                                            _ptr(tgt), _ptr(tgt_sfc), _ptr(yto), _ptr(yto_sfc), _ptr(x_raw), _ptr(x_sfc_n), _ptr(sc),
                                            _ptr(d_p), _ptr(d_s), self._stream())
        if rc != 0:
            raise RuntimeError(f"csa_phys_train_loss failed ({rc}): {_lib.last_error()}")
        names = ("loss", "huber", "mse", "mae", "energy", "water", "precip_sum_mse")
        return (dict(zip(names, sc.tolist())) if scalars else sc), d_p, d_s       # (scalars=False: the device tensor, no host sync)

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """torch.optim.Adam (the reference's default, train_rnn_rollout_torchscript_hydra.py:678); weight_decay is its L2 term."""
        rc = _lib.lib().csa_phys_train_adam_step(self._h, _ptr(self.grads), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                 float(weight_decay), self._stream())
        if rc != 0:
            raise RuntimeError(f"csa_phys_train_adam_step failed ({rc}): {_lib.last_error()}")


_W_ORDER = (["hyam", "hybm", "hyai", "hybi", "yscale_lev", "yscale_sca", "xdiv_sca", "xmean_sca",
             "mlp_initial.weight", "mlp_initial.bias", "mlp_surface1.weight", "mlp_surface1.bias"]
            + [f"rnn{r}.{n}_l0" for r in (1, 2) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
            + [f"{m}.{n}" for m in ("mlp_latent", "mlp_output", "mlp_precip_release") for n in ("weight", "bias")])
_W_LW = ["gas_optics_model_lw.xmin", "gas_optics_model_lw.xdiv", "gas_optics_model_lw.ymean", "gas_optics_model_lw.ystd"] + \
    [f"gas_optics_model_lw.mlp{i}.{n}" for i in (1, 2, 3) for n in ("weight", "bias")]


class physical_RNN_wrapped(torch.nn.Module):
    """Host mirror of the reference's DEPLOYED physRNN modules: the frozen exports rnn/saved_models/physRNN_physRad-*_nx21_*_wrapped.pt,
    i.e. rnn/utils.py::model_wrapper (:72-295) around the nx21 generation of rnn/models/models_phys.py::physical_RNN_autoreg.  Same
    call as the export, raw physical units in and out:

        forward(x_main0 (B,60,20), x_sfc0 (B,19), rnn1_mem (50,B,16)) -> (out_lev (B,60,6), out_sfc (B,8), rnn1_mem (50,B,16))

    `state_dict`: the export's constants under the names they had before freezing; `cfg`: the variant's switches and maps
    (`rad_updated_qv`, `n_ir`, `n_mix_end`, `band_idx`) -- climsim_amd/frozen_extract.py recovers both from an export's serialised
    code.  The export draws rnn2's initial state, the stochastic third RNN's state and noise and the SW humidity coin inside forward;
    pass `hx2=`, `hx1=`, `eps3=`, `mask_u=` to make a call reproducible (whatever is omitted is drawn here on the device)."""

    def __init__(self, state_dict, cfg, *, max_batch=4096):
        super().__init__()
        self._h = None
        if not torch.cuda.is_available():
            raise RuntimeError("climsim_amd needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        f = lambda k: np.ascontiguousarray(state_dict[k].detach().cpu().numpy() if isinstance(state_dict[k], torch.Tensor) else state_dict[k], np.float32)
        nreg = int(f("mlp_qv_crm.weight").shape[0])
        self.nh, self.ng = int(f("mlp_initial.weight").shape[0]), nreg
        self.stochastic, liq_head = "rnn3.weight_ih" in state_dict, "mlp_liq_frac_crm.weight" in state_dict

        def pad_rows(a, fill=0.0):          # per-region / per-g-point axis -> 16
            out = np.full((16,) + a.shape[1:], fill, np.float32)
            out[:a.shape[0]] = a
            return out
        arrs = [f(k) for k in _W_ORDER]
        # decoder sub-generation without a sub-grid temperature (num27378 / num45826 / num74834: the physRad decoder): no mlp_t_crm
        # (zero head), ONE eddy diffusivity per level (its row for every region), and -- with the clear-sky region -- a zero first row
        # in front of the nreg - 1 rows of the condensate heads
        grid_T = "mlp_t_crm.weight" not in state_dict
        clear0 = grid_T and f("mlp_qn_crm.weight").shape[0] == nreg - 1
        for m in _HEADS:
            if grid_T and m == "mlp_t_crm":
                arrs += [np.zeros((16, self.nh), np.float32), np.zeros(16, np.float32)]
                continue
            w_, b_ = f(m + ".weight"), f(m + ".bias")
            if grid_T and m == "mlp_eddy_diff":
                w_, b_ = np.repeat(w_, nreg, 0), np.repeat(b_, nreg, 0)
            if clear0 and m in ("mlp_qn_crm", "mlp_evap_cond_vapor_crm"):
                w_, b_ = np.concatenate([np.zeros((1, self.nh), np.float32), w_]), np.concatenate([np.zeros(1, np.float32), b_])
            arrs += [pad_rows(w_), pad_rows(b_, -1.0e30 if m == "mlp_subgrid_area_frac" else 0.0)]
        arrs += [f("yscale_sca_rad"), pad_rows(f("solar_weights").reshape(-1))] + [f(k) for k in _W_LW]
        arrs += [pad_rows(f("gas_optics_lw_reduce1.weight")), pad_rows(f("gas_optics_lw_reduce1.bias")),
                 pad_rows(f("gas_optics_lw_reduce2.weight")), pad_rows(f("gas_optics_lw_reduce2.bias"), -1.0e30)]
        sw_head = "mlp_sw_optprops1.weight" in state_dict      # earlier sub-generation: SW optical properties from one two-layer MLP
        sw_e3sm = "gas_optics_model_sw1.ystd" in state_dict
        self.ngk = 0
        if sw_head:       # 24 -> nhid <= 32 -> 3 x nreg; hidden units and g-points zero-padded to 32 / 16
            w1, b1, w2, b2 = (f("mlp_sw_optprops" + k) for k in ("1.weight", "1.bias", "2.weight", "2.bias"))
            nhid = w1.shape[0]
            if w1.shape[1] != 24 or nhid > 32 or w2.shape != (3 * nreg, nhid):
                raise RuntimeError("physRNN (frozen export): the SW head is built for 24 -> (<= 32) -> 3 x nreg")
            W1, B1, W2, B2 = np.zeros((32, 24), np.float32), np.zeros(32, np.float32), np.zeros((48, 32), np.float32), np.zeros(48, np.float32)
            W1[:nhid], B1[:nhid] = w1, b1
            for c in range(3):
                W2[16 * c:16 * c + nreg, :nhid], B2[16 * c:16 * c + nreg] = w2[nreg * c:nreg * (c + 1)], b2[nreg * c:nreg * (c + 1)]
            arrs += [W1, B1, W2, B2, np.ascontiguousarray(f("lbd_qn").reshape(-1))]
        elif sw_e3sm:      # the unfrozen physics_rad_e3sm form of the SW gas optics (112 k-points, mean of the two humidity variants)
            arrs.append(_sw_gas_block(state_dict, nreg))
            self._cloud_table(arrs, f, state_dict, cfg, nreg, pad_rows)
        else:
            self._sw_gas_arrays(arrs, f, state_dict, cfg, nreg, pad_rows)
        band_matrix = "cloud_band_to_gpt" in state_dict
        mix = (float(f("mix_near").reshape(-1)[0]), float(f("mix_vis").reshape(-1)[0])) if "mix_near" in state_dict else (0.5, 0.5)
        bits = (1 if cfg.get("sfc_sw_down") else 0) | (2 if cfg.get("cld_liq_from_updated_T") else 0) | (4 if cfg.get("rad_updated_qn") else 0) | \
            (8 if grid_T else 0) | (16 if clear0 else 0) | (0 if cfg.get("cld_qn_updated", True) else 32) | \
            (0 if cfg.get("rad_updated_T", True) else 64) | (128 if cfg.get("rnn3_last_mul") else 0)
        arrs.append(np.asarray([cfg["n_ir"], cfg["n_mix_end"], mix[0], mix[1], self.ngk, int(bool(cfg.get("ice_optics_on_ice_radius"))),
                                int(band_matrix), bits], np.float32))
        arrs += [f("xmean_lev"), f("xdiv_lev"), f("lbd_qc"), f("lbd_qi")]
        flags = (2 if liq_head else 0) | (4 if self.stochastic else 0) | (256 if cfg.get("rad_updated_qv") else 0) | (512 if sw_head else 0) | (64 if sw_e3sm else 0)
        if liq_head:
            arrs += [pad_rows(f("mlp_liq_frac_crm.weight")), pad_rows(f("mlp_liq_frac_crm.bias"))]
        if self.stochastic:
            arrs += [f("rnn3.weight_ih"), f("rnn3.weight_zh"), f("rnn3.weight_encoder")]
        FP = ctypes.POINTER(ctypes.c_float)
        warr = (FP * len(arrs))(*[a.ctypes.data_as(FP) for a in arrs])
        h = ctypes.c_void_p()
        rc = _lib.lib().csa_phys_wrapped_create(self.nh, nreg, flags, warr, int(max_batch), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_phys_wrapped_create failed ({rc}): {_lib.last_error()}")
        self._h, self.max_batch = h, max_batch

    @classmethod
    def from_export(cls, path, *, max_batch=4096):
        """Drop-in for `torch.jit.load(path)` of a frozen `*_wrapped.pt` export: reads the graph constants and the switches of the
        export's code variant (climsim_amd/frozen_extract.py) and builds the HIP model with them."""
        from .frozen_extract import load_export
        state_dict, cfg = load_export(path)
        return cls(state_dict, cfg, max_batch=max_batch)

    def _sw_gas_arrays(self, arrs, f, state_dict, cfg, nreg, pad_rows):
        """SW gas-optics block (csrc/phys.h SWX_*) and the cloud-optics table of the later sub-generations"""
        pad8 = lambda v, fill: np.concatenate([v.reshape(-1), np.full(8 - v.size, fill, np.float32)])
        blk = [pad8(f("gas_optics_model_sw1.xmin"), 0.0), pad8(f("gas_optics_model_sw1.xdiv"), 1.0)]
        for m in ("gas_optics_model_sw1", "gas_optics_model_sw2"):
            w1 = f(m + ".mlp1.weight")
            ngk = int(f(m + ".mlp3.weight").shape[0])
            if w1.shape != (32, 7) or f(m + ".mlp2.weight").shape != (32, 32) or f(m + ".mlp3.weight").shape[1] != 32 or \
                    (ngk != nreg and "gas_optics_sw_reduce1.weight" not in state_dict) or ngk > 16:
                raise RuntimeError("physRNN (frozen export): built for the shipped 7 -> 32 -> 32 -> ng SW gas-optics models")
            blk += [np.concatenate([w1, np.zeros((32, 1), np.float32)], 1).ravel(), f(m + ".mlp1.bias"), f(m + ".mlp2.weight").ravel(),
                    f(m + ".mlp2.bias"), pad_rows(f(m + ".mlp3.weight")).ravel(), pad_rows(f(m + ".mlp3.bias"))]
        # k-point -> g-point reductions behind the humidity coin (sub-generation of num11916 / num87824), zero block otherwise
        self.ngk = 0
        for r in ("gas_optics_sw_reduce1", "gas_optics_sw_reduce2"):
            wr, br = np.zeros((16, 16), np.float32), np.zeros(16, np.float32)
            if r + ".weight" in state_dict:
                w0 = f(r + ".weight")
                if nreg != 16 or w0.shape[0] != 16:
                    raise RuntimeError("physRNN (frozen export): the SW k-point reduction is built for 16 g-points")
                self.ngk = int(w0.shape[1])
                wr[:, :self.ngk], br[:] = w0, f(r + ".bias")
            blk += [wr.ravel(), br]
        arrs.append(np.ascontiguousarray(np.concatenate(blk), np.float32))
        self._cloud_table(arrs, f, state_dict, cfg, nreg, pad_rows)

    def _cloud_table(self, arrs, f, state_dict, cfg, nreg, pad_rows):
        tab = np.asarray(_SLINGO + _EBERT_CURRY, np.float32)
        band_matrix = "cloud_band_to_gpt" in state_dict
        if band_matrix:           # four-band tables + the learned (4, ng) band -> g-point matrix
            arrs.append(np.ascontiguousarray(np.concatenate([tab.ravel(), pad_rows(f("cloud_band_to_gpt").T).T.ravel()]), np.float32))
        else:
            arrs.append(np.ascontiguousarray(tab[:, list(cfg["band_idx"]) + [0] * (16 - nreg)]))

    def forward(self, x_main0, x_sfc0, rnn1_mem, hx2=None, hx1=None, eps3=None, mask_u=None, _srnn=None):
        B = x_main0.shape[0]
        x_main0 = _check(x_main0, (B, 60, 20), "x_main0")
        x_sfc0 = _check(x_sfc0, (B, 19), "x_sfc0")
        rnn1_mem = _check(rnn1_mem, (50, B, 16), "rnn1_mem")
        dev, nh = self.device, self.nh
        hx2 = torch.randn(B, nh, device=dev) if hx2 is None else _check(hx2, (B, nh), "hx2")
        if self.stochastic:
            hx1 = torch.randn(B, nh, device=dev) if hx1 is None else _check(hx1, (B, nh), "hx1")
            eps3 = torch.randn(50, B, nh, device=dev) if eps3 is None else _check(eps3, (50, B, nh), "eps3")
        else:
            hx1 = eps3 = None
        nm = self.ngk or self.ng          # the coin is drawn per k-point of the SW gas models (= g-points unless the export reduces them)
        mask_u = torch.rand(60, B, nm, device=dev) if mask_u is None else _check(mask_u, (60, B, nm), "mask_u")
        if _srnn is not None:
            _srnn = _check(_srnn, (50, B, nh), "srnn")
        out_lev, out_sfc, mem_out = torch.empty(B, 60, 6, device=dev), torch.empty(B, 8, device=dev), torch.empty(50, B, 16, device=dev)
        rc = _lib.lib().csa_phys_wrapped_forward(self._h, B, _ptr(x_main0), _ptr(x_sfc0), _ptr(rnn1_mem), _ptr(hx2), _ptr(hx1), _ptr(eps3),
                                                 _ptr(mask_u), _ptr(_srnn), _ptr(out_lev), _ptr(out_sfc), _ptr(mem_out),
                                                 ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"csa_phys_wrapped_forward failed ({rc}): {_lib.last_error()}")
        return out_lev, out_sfc, mem_out

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().csa_phys_destroy(self._h)
                self._h = None
        except Exception:
            pass
