"""Device-side twin of the reference's training-data generator (rnn/utils.py:1870-2371, class generator_xy).

The reference reads a chunk of time steps from HDF5 and then normalises inputs and builds targets with numpy/numba
on host cores inside DataLoader workers.  Here the chunk (whatever array-like the caller has: numpy, memmap or an
HDF5 dataset object -- file I/O is not part of this package) is copied to the GPU once and ONE HIP kernel produces
the seven tensors of `__getitem__`.  Same constructor keywords and return order as the reference class."""
import ctypes

import numpy as np
import torch

from . import _lib

_FP = ctypes.POINTER(ctypes.c_float)
_CLD = {"none": 0, "exp": 1, "sqrt": 2}


class generator_xy:
    def __init__(self, data, nloc=384, xcoeffs=None, ycoeffs=None, xcoeffs_ref=None, ycoeffs_ref=None,
                 lbd_qc=None, lbd_qi=None, lbd_qn=None, v4_to_v5_inputs=False, cld_inp_transformation="exp",
                 remove_past_sfc_inputs=False, qinput_prune=False, rh_input_to_q=False, include_q_input=False,
                 rh_prune=False, output_prune=False, mp_mode=0, hybm=None, hyam=None, include_prev_inputs=False,
                 include_prev_outputs=False, snowhice_fix=True, device=None):
        self._h = None
        if not torch.cuda.is_available():
            raise RuntimeError("climsim_amd needs a HIP device: the product path has no CPU fallback")
        self.include_prev_inputs, self.include_prev_outputs = bool(include_prev_inputs), bool(include_prev_outputs)
        if cld_inp_transformation not in _CLD:
            raise NotImplementedError()
        if ycoeffs is None:
            raise RuntimeError("ycoeffs (yscale_lev, yscale_sca) are required")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.data = data                       # mapping with input_lev, input_sca, output_lev, output_sca
        dims = data["input_lev"].shape
        if len(dims) != 4:
            raise NotImplementedError("expected input_lev of shape (ntime, nloc, nlev, nx)")
        self.ntimesteps, self.nloc, self.nlev, nx_in = dims
        self.ncol = self.nloc
        # rnn/utils.py:2012-2023: the previous step's outputs (5) / inputs (6) are appended as extra level inputs and the
        # first time step is lost to them
        nx_in += (5 if include_prev_outputs else 0) + (6 if include_prev_inputs else 0)
        if include_prev_inputs or include_prev_outputs:
            self.ntimesteps -= 1
        cfg = _lib.CsaGenConfig()
        cfg.nlev, cfg.nx_in = self.nlev, nx_in
        cfg.nx_sfc_in = data["input_sca"].shape[-1]
        cfg.ny_sfc = data["output_sca"].shape[-1]
        cfg.remove_past_sfc_inputs, cfg.snowhice_fix = int(remove_past_sfc_inputs), int(snowhice_fix)
        cfg.rh_prune, cfg.qinput_prune, cfg.output_prune = int(rh_prune), int(qinput_prune), int(output_prune)
        # rnn/utils.py:2183-2194: the conversion runs when rh_input_to_q is set; include_q_input then selects append vs replace and
        # does nothing on its own (train_rnn_rollout_torchscript_hydra.py:231-233 forces rh_input_to_q when it is configured)
        cfg.q_mode = 0 if not rh_input_to_q else (1 if include_q_input else 2)
        if rh_input_to_q and (hyam is None or hybm is None):
            raise NotImplementedError("Please provide hyam,hybm")
        cfg.cld_inp_transformation = _CLD[cld_inp_transformation]
        cfg.v4_to_v5_inputs = int(v4_to_v5_inputs)
        cfg.apply_new_input_scaling = int(xcoeffs is not None)
        cfg.reverse_input_norm, cfg.reverse_output_norm = int(xcoeffs_ref is not None), int(ycoeffs_ref is not None)
        cfg.mp_mode = int(mp_mode)
        self.cfg = cfg
        keep = {}

        def arr(a):
            return None if a is None else np.ascontiguousarray(a, np.float32)
        if xcoeffs is not None:
            keep["xmean_lev"], keep["xdiv_lev"] = arr(xcoeffs[0][0]), arr(xcoeffs[0][1])
            keep["xmean_sca"], keep["xdiv_sca"] = arr(xcoeffs[1][0]), arr(xcoeffs[1][1])
        keep["yscale_lev"], keep["yscale_sca"] = arr(ycoeffs[0]), arr(ycoeffs[1])
        for k, v in (("lbd_qc", lbd_qc), ("lbd_qi", lbd_qi), ("lbd_qn", lbd_qn), ("hyam", hyam), ("hybm", hybm)):
            keep[k] = arr(v)
        if xcoeffs_ref is not None:
            keep["xref_mean"], keep["xref_div"] = arr(xcoeffs_ref[0][0]), arr(xcoeffs_ref[0][1])
            keep["xsref_mean"], keep["xsref_div"] = arr(xcoeffs_ref[1][0]), arr(xcoeffs_ref[1][1])
        if ycoeffs_ref is not None:
            keep["yref_lev"], keep["yref_sca"] = arr(ycoeffs_ref[0]), arr(ycoeffs_ref[1])
        co = _lib.CsaGenCoeffs()
        for k, v in keep.items():
            if v is not None:
                setattr(co, k, v.ctypes.data_as(_FP))
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = _lib.lib().csa_gen_create(ctypes.byref(cfg), ctypes.byref(co), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"csa_gen_create failed ({rc}): {_lib.last_error()}")
        self._h = h
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.lib().csa_gen_dims(h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        # widths of the tensors __getitem__ returns; the public nx / nx_sfc / ny / ny_sfc are, as in the reference class
        # (rnn/utils.py:2003-2026), the widths of the stored datasets (nx with the previous-step columns added)
        self.nx_out, self.nx_sfc_out, self.ny_out = a.value, b.value, c.value
        self.nx, self.nx_sfc, self.ny, self.ny_sfc = nx_in, cfg.nx_sfc_in, data["output_lev"].shape[-1], cfg.ny_sfc

    def __len__(self):
        return self.ntimesteps * self.ncol

    def _dev(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(self.device, torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device, non_blocking=True)

    def prepare(self, x_lev_b, x_sfc_b, y_lev_b, y_sfc_b):
        """The arithmetic of __getitem__ on an already loaded chunk; arrays may be host or device, any leading shape."""
        c = self.cfg
        xl = self._dev(x_lev_b).reshape(-1, c.nlev, c.nx_in)
        N = xl.shape[0]
        xs = self._dev(x_sfc_b).reshape(N, c.nx_sfc_in)
        yl = self._dev(y_lev_b).reshape(N, c.nlev, 6)
        ys = self._dev(y_sfc_b).reshape(N, c.ny_sfc)
        e = lambda *s: torch.empty(*s, device=self.device)
        out = (e(N, c.nlev, self.nx_out), e(N, self.nx_sfc_out), e(N, c.nlev, self.ny_out), e(N, c.ny_sfc), e(N, c.nlev, self.nx_out),
               e(N, c.nlev, 6), e(N, c.ny_sfc))
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        rc = _lib.lib().csa_gen_batch(self._h, N, P(xl), P(xs), P(yl), P(ys), *[P(t) for t in out],
                                      ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"csa_gen_batch failed ({rc}): {_lib.last_error()}")
        return out

    def __getitem__(self, indices):
        d = self.data
        idx = list(indices)
        x_lev = self._dev(d["input_lev"][idx])
        if self.include_prev_inputs or self.include_prev_outputs:      # rnn/utils.py:2242-2249, 2262-2279, 2291-2297
            if idx[0] <= 0:
                raise NotImplementedError("First time index cannot be zero as it's used for memory")
            prev = [idx[0] - 1] + idx[:-1]
            if self.include_prev_outputs:
                x_lev = torch.cat((x_lev, self._dev(d["output_lev"][prev])[..., 0:5]), dim=-1)
            if self.include_prev_inputs:
                x_lev = torch.cat((x_lev, self._dev(d["input_lev"][prev])[..., 0:6]), dim=-1)
        return self.prepare(x_lev, d["input_sca"][idx], d["output_lev"][idx], d["output_sca"][idx])

    def __del__(self):
        try:
            if self._h is not None:
                _lib.lib().csa_gen_destroy(self._h)
                self._h = None
        except Exception:
            pass
