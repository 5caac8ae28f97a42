"""Host-side mirrors of the reference's deployment interface.

Same class / method names, argument meaning and error behaviour (torch RuntimeError on
shape / dtype / device mismatch) as the reference's TorchScript wrappers, so that a harness
written for `torch.jit.load("v4_rnn-memory_wrapper_constrained_huber.pt")` runs unchanged:

    NewModel_constraint.forward(x_main, x_sfc)                  rnn/save_wrapper.py:255-298
    NewModel_constraint.forward(x_main, x_sfc, rnn1_mem)        rnn/save_wrapper_mem.py:499-545
    model_wrapper.forward(x_main0, x_sfc0, rnn1_mem)            rnn/utils.py:260-295
    RNN_autoreg.forward(inp_list) / .postprocessing(...)        rnn/models/models.py:432-608, 273-339

All arithmetic happens in the HIP kernels behind the C ABI; these classes only validate
tensors, own the handle and pass device pointers + the current stream.  Recurrent state is
owned by the caller, exactly as in the reference (rollout harness save_wrapper_mem.py:827-852).
"""
import numpy as np
import torch
import torch.nn as nn

from .emulator import Emulator


def _load_npz(path):
    d = np.load(path)
    consts = {k[2:]: d[k] for k in d.files if k.startswith("c.")}
    weights = {k[2:]: d[k] for k in d.files if k.startswith("w.")}
    flags = {k[6:]: int(d[k]) for k in d.files if k.startswith("flags.")}
    return consts, weights, flags


class NewModel_constraint(nn.Module):
    """Packed v4 wrapper (stateless or stateful), legacy generation = the shipped artefacts.

    forward(x_main (B,60,15), x_sfc (B,19)[, rnn1_mem (B,60,16)]) -> yout (B, 368[+960]) with
    yout[:, 0:360] = dT,dqv,dqliq,dqice,du,dv (60 each), [360:368] surface, [368:] new memory.
    The artefacts draw hx2,cx2 ~ N(0,1) inside forward; pass `noise=(hx2,cx2)` to fix them.
    """

    def __init__(self, consts, state_dict, *, scrub_out_nan=None, snowhice_fix=False, qinput_prune=False,
                 rh_prune=False, scrub_inf=False, max_batch=4096):
        super().__init__()
        stateful = "mlp_latent.weight" in state_dict
        if scrub_out_nan is None:
            scrub_out_nan = False   # neither shipped artefact scrubs its output (TorchScript code)
        self.emulator = Emulator(consts, state_dict, legacy=True, use_lstm=True, mp_mode=1,
                                 snowhice_fix=snowhice_fix, qinput_prune=qinput_prune, rh_prune=rh_prune,
                                 scrub_inf=scrub_inf, scrub_out_nan=scrub_out_nan, max_batch=max_batch)
        self.stateful = stateful
        self.nmem = self.emulator.cfg.nh_mem

    @classmethod
    def from_npz(cls, path, **kw):
        consts, weights, _ = _load_npz(path)
        return cls(consts, weights, **kw)

    def forward(self, x_main, x_sfc, rnn1_mem=None, noise=None):
        if self.stateful and rnn1_mem is None:
            raise RuntimeError("forward() is missing value for argument 'rnn1_mem'")
        if not self.stateful and rnn1_mem is not None:
            raise RuntimeError("forward() expected at most 2 tensor arguments for the stateless wrapper")
        hx2, cx2 = (None, None) if noise is None else noise
        return self.emulator.forward_packed(x_main, x_sfc, rnn1_mem, hx2, cx2)


class RNN_autoreg(nn.Module):
    """Model-level interface of the current generation (normalised inputs in, normalised outputs out).

    forward([x_main_norm (B,60,nx), x_sfc_norm (B,19), rnn_mem (60,B,nh_mem)]) ->
        (out (B,60,ny), out_sfc (B,8), rnn_mem (60,B,nh_mem))
    """

    def __init__(self, consts, state_dict, *, use_lstm=True, output_prune=False, mp_mode=1, max_batch=4096):
        super().__init__()
        self.emulator = Emulator(consts, state_dict, legacy=False, use_lstm=use_lstm, mp_mode=mp_mode,
                                 output_prune=output_prune, scrub_inf=True, max_batch=max_batch)
        c = self.emulator.cfg
        self.nlev, self.nlev_mem, self.nh_mem, self.ny, self.ny_sfc = c.nlev, c.nlev, c.nh_mem, c.ny, c.ny_sfc
        self.mp_mode = mp_mode
        dev = self.emulator.device
        for k in ("yscale_lev", "yscale_sca", "xmean_lev", "xdiv_lev", "xmean_sca", "xdiv_sca", "hyam", "hybm",
                  "lbd_qc", "lbd_qi"):
            self.register_buffer(k, torch.from_numpy(np.ascontiguousarray(consts[k], np.float32)).to(dev))

    def forward(self, inp_list, noise=None):
        """noise = (hx0, cx0, eps) for add_stochastic_layer models; drawn with torch.randn when omitted, as the
        reference does inside forward (models.py:466-468)."""
        x_main, x_sfc, rnn_mem = inp_list[0], inp_list[1], inp_list[2]
        return self.emulator.model_forward(x_main, x_sfc, rnn_mem, noise=noise)

    def postprocessing(self, out, out_sfc, x_denorm):
        """models.py:273-339: de-normalise and (mp_mode != 0) partition the cloud-water tendency: (B,60,ny), (B,8), raw inputs
        (B,60,nx) -> (B,60,6) [dT,dqv,dqliq,dqice,du,dv], (B,8).  mp_mode 0 returns its arguments unchanged, as upstream."""
        return self.emulator.postprocess(out, out_sfc, x_denorm)


class model_wrapper(nn.Module):
    """Tuple ("ftorch") wrapper of the current generation, rnn/utils.py:72-295 (v4 inputs, or v5_input=True: :186-198).

    forward(x_main0 (B,60,15), x_sfc0 (B,19), rnn1_mem (60,B,nh_mem)) ->
        (out_lev (B,60,6), out_sfc (B,8), rnn1_mem (60,B,nh_mem))
    """

    def __init__(self, consts, state_dict, *, use_lstm=True, output_prune=False, mp_mode=1,
                 qinput_prune=False, rh_prune=False, snowhice_fix=True, rh_to_q=False, include_q_input=None,
                 v5_input=False, max_batch=4096):
        super().__init__()
        nx = np.asarray(consts["xmean_lev"]).shape[1]
        if include_q_input is None:          # rnn/utils.py:107-111: nx in [16, 21] means q was appended
            include_q_input = nx in (16, 21)
        q_mode = 1 if include_q_input else (2 if rh_to_q else 0)
        self.emulator = Emulator(consts, state_dict, legacy=False, use_lstm=use_lstm, mp_mode=mp_mode,
                                 output_prune=output_prune, snowhice_fix=snowhice_fix,
                                 qinput_prune=qinput_prune, rh_prune=rh_prune, scrub_inf=True,
                                 q_input_mode=q_mode, v5_input=v5_input, max_batch=max_batch)
        c = self.emulator.cfg
        self.nx, self.nmem, self.nlev_mem = c.nx, c.nh_mem, c.nlev

    def forward(self, x_main0, x_sfc0, rnn1_mem, noise=None):
        return self.emulator.forward_tuple(x_main0, x_sfc0, rnn1_mem, noise=noise)


class NewModel_constraint_ar(nn.Module):
    """Stateful + AR-noise packed wrapper, rnn/save_wrapper_mem.py:682-727 (third row of SURVEY section 8b):

        forward(x_main (B,60,15), x_sfc (B,19), rnn1_mem (B,60,nh_mem), eps_prev (B,60,nh)) -> (B, 368 + 60*nh_mem + 60*nh)

    around the stochastic current-generation model (add_stochastic_layer).  The caller re-slices yout[:, 368:368+960] (memory)
    and yout[:, 368+960:] (eps) and feeds both back.  The eps block is the eps that was used: the AR(1) model classes that
    updated it are commented out upstream, so there is no update rule to restate."""

    def __init__(self, consts, state_dict, *, output_prune=False, qinput_prune=False, rh_prune=False, snowhice_fix=True, max_batch=4096):
        super().__init__()
        if "rnn2.weight_encoder" not in state_dict:
            raise RuntimeError("the AR-noise wrapper wraps a stochastic model (rnn2.weight_encoder missing)")
        self.emulator = Emulator(consts, state_dict, legacy=False, use_lstm=True, mp_mode=1, output_prune=output_prune,
                                 snowhice_fix=snowhice_fix, qinput_prune=qinput_prune, rh_prune=rh_prune, scrub_inf=True,
                                 max_batch=max_batch)
        self.nmem = self.emulator.cfg.nh_mem

    def forward(self, x_main, x_sfc, rnn1_mem, eps_prev, noise=None):
        hx0, cx0 = (None, None) if noise is None else noise
        return self.emulator.forward_packed_noise(x_main, x_sfc, rnn1_mem, eps_prev, hx0, cx0)
