"""ctypes binding of libclimsim_amd.so (the C ABI in include/climsim_amd.h).

The product path has NO CPU fallback: if the HIP library is missing or cannot be loaded the
import of anything that computes raises, loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSA_LIB_PATH: a diagnostic build of the same sources (e.g. tools/build_exact_gates.sh); never a different implementation
LIB_PATH = os.environ.get("CSA_LIB_PATH") or os.path.join(_HERE, "libclimsim_amd.so")

_F = ctypes.c_void_p   # device pointers travel as integers

CONFIG_FIELDS = ["nlev", "nx", "nx_sfc", "ny", "ny_sfc", "nh1", "nh2", "nh_mem", "use_lstm", "legacy",
                 "output_prune", "mp_mode", "snowhice_fix", "qinput_prune", "rh_prune", "scrub_inf",
                 "scrub_out_nan", "q_input_mode", "add_stochastic_layer", "v5_input"]
PARAM_FIELDS = ["xmean_lev", "xdiv_lev", "xmean_sca", "xdiv_sca", "lbd_qc", "lbd_qi",
                "yscale_lev", "yscale_sca", "hyam", "hybm",
                "mlp_initial_w", "mlp_initial_b", "mlp_surface1_w", "mlp_surface1_b",
                "mlp_surface2_w", "mlp_surface2_b", "mlp_toa1_w", "mlp_toa1_b", "mlp_toa2_w", "mlp_toa2_b",
                "rnn1_w_ih", "rnn1_w_hh", "rnn1_b_ih", "rnn1_b_hh",
                "rnn2_w_ih", "rnn2_w_hh", "rnn2_b_ih", "rnn2_b_hh",
                "mlp_latent_w", "mlp_latent_b", "mlp_output_w", "mlp_output_b",
                "mlp_surface_output_w", "mlp_surface_output_b",
                "rnn0_w_ih", "rnn0_w_hh", "rnn0_b_ih", "rnn0_b_hh", "rnn2_weight_encoder", "lbd_qn"]

# Every symbol include/climsim_amd.h declares (checked by tests/test_abi.py).
SYMBOLS = ["csa_create", "csa_destroy", "csa_set_params", "csa_packed_width", "csa_max_batch",
           "csa_forward_packed", "csa_forward_tuple", "csa_model_forward", "csa_forward_tuple_noise",
           "csa_model_forward_noise", "csa_forward_packed_noise", "csa_postprocess", "csa_tap_rnn1",
           "csa_tap_rnn2", "csa_last_error", "csa_version", "csa_set_profiling", "csa_reset_profile",
           "csa_get_profile", "csa_stage_name", "csa_set_halves", "csa_set_rec1_max_batch", "csa_set_small_gemm_rows", "csa_set_gemm_split", "csa_debug_stage",
           "csa_train_create", "csa_train_destroy", "csa_train_num_params", "csa_train_num_tensors",
           "csa_train_param_info", "csa_train_params", "csa_train_sync_params", "csa_train_copy_state", "csa_train_forward", "csa_train_forward_noise",
           "csa_train_backward", "csa_train_set_deferred", "csa_train_flush_wgrad", "csa_train_loss", "csa_train_adam",
           "csa_train_set_profiling", "csa_train_reset_profile", "csa_train_get_profile", "csa_train_stage_name",
           "csa_mlp_create", "csa_mlp_destroy", "csa_mlp_forward",
           "csa_mlp_train_create", "csa_mlp_train_destroy", "csa_mlp_train_num_params", "csa_mlp_train_copy_params",
           "csa_mlp_train_forward", "csa_mlp_train_backward", "csa_mlp_train_adam",
           "csa_cnn_create", "csa_cnn_destroy", "csa_cnn_forward", "csa_cnn_reshape_to", "csa_cnn_reshape_from",
           "csa_cnn_train_create", "csa_cnn_train_destroy", "csa_cnn_train_num_params", "csa_cnn_train_num_layers",
           "csa_cnn_train_params", "csa_cnn_train_get_params", "csa_cnn_train_set_params", "csa_cnn_train_get_act", "csa_cnn_train_layer_info", "csa_cnn_train_forward", "csa_cnn_train_backward",
           "csa_cnn_train_adam",
           "csa_gen_create", "csa_gen_destroy", "csa_gen_dims", "csa_gen_batch",
           "csa_crps", "csa_crps_backward", "csa_spread_skill",
           "csa_eval_scratch_bytes", "csa_eval_metrics", "csa_eval_crps", "csa_derive_inputs",
           "csa_phys_create", "csa_phys_destroy", "csa_phys_forward", "csa_phys_tap", "csa_phys_rad_create", "csa_phys_postprocess", "csa_phys_forward_noise", "csa_phys_debug_forward_srnn", "csa_phys_debug_rnn3", "csa_phys_wrapped_create", "csa_phys_wrapped_forward",
           "csa_phys_train_enable", "csa_phys_train_num_params", "csa_phys_train_param_info", "csa_phys_train_get_params", "csa_phys_train_set_params",
           "csa_phys_train_forward", "csa_phys_train_backward", "csa_phys_train_adam_step", "csa_phys_train_loss",
           "csa_online_create", "csa_online_destroy", "csa_online_dims", "csa_online_forward",
           "csa_stoch_gru5_create", "csa_stoch_lstm4_create", "csa_stoch_destroy", "csa_stoch_gru5_forward",
           "csa_stoch_lstm4_forward", "csa_stoch_enable_training", "csa_stoch_num_params", "csa_stoch_gru5_forward_train",
           "csa_stoch_lstm4_forward_train", "csa_stoch_gru5_backward", "csa_stoch_lstm4_backward", "csa_stoch_activation_floats",
           "csa_stoch_set_activations"]


class CsaConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in CONFIG_FIELDS]


GEN_CONFIG_FIELDS = ["nlev", "nx_in", "nx_sfc_in", "ny_sfc", "remove_past_sfc_inputs", "snowhice_fix", "rh_prune",
                     "qinput_prune", "output_prune", "q_mode", "cld_inp_transformation", "v4_to_v5_inputs",
                     "apply_new_input_scaling", "reverse_input_norm", "reverse_output_norm", "mp_mode"]
GEN_COEFF_FIELDS = ["xmean_lev", "xdiv_lev", "xmean_sca", "xdiv_sca", "yscale_lev", "yscale_sca", "lbd_qc", "lbd_qi", "lbd_qn",
                    "hyam", "hybm", "xref_mean", "xref_div", "xsref_mean", "xsref_div", "yref_lev", "yref_sca"]


class CsaGenConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in GEN_CONFIG_FIELDS]


class CsaGenCoeffs(ctypes.Structure):
    _fields_ = [(n, ctypes.POINTER(ctypes.c_float)) for n in GEN_COEFF_FIELDS]


class CsaParams(ctypes.Structure):
    _fields_ = [(n, ctypes.POINTER(ctypes.c_float)) for n in PARAM_FIELDS]


_lib = None


def lib():
    """Load the HIP library; raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m climsim_amd.build` "
            "(hipcc --offload-arch=gfx950). climsim_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    H = ctypes.c_void_p
    i = ctypes.c_int
    L.csa_create.argtypes = [ctypes.POINTER(CsaConfig), ctypes.POINTER(CsaParams), i, ctypes.POINTER(H)]
    L.csa_destroy.argtypes = [H]
    L.csa_set_params.argtypes = [H, ctypes.POINTER(CsaParams)]
    L.csa_packed_width.argtypes = [H]
    L.csa_max_batch.argtypes = [H]
    L.csa_forward_packed.argtypes = [H, i, _F, _F, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_forward_tuple.argtypes = [H, i, _F, _F, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_model_forward.argtypes = [H, i, _F, _F, _F, _F, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_forward_tuple_noise.argtypes = [H, i] + [_F] * 9 + [ctypes.c_void_p]
    L.csa_model_forward_noise.argtypes = [H, i] + [_F] * 9 + [ctypes.c_void_p]
    L.csa_forward_packed_noise.argtypes = [H, i] + [_F] * 7 + [ctypes.c_void_p]
    L.csa_postprocess.argtypes = [H, i, _F, _F, _F, i, _F, _F, ctypes.c_void_p]
    L.csa_tap_rnn1.argtypes = [H]
    L.csa_tap_rnn1.restype = ctypes.c_void_p
    L.csa_tap_rnn2.argtypes = [H]
    L.csa_tap_rnn2.restype = ctypes.c_void_p
    L.csa_set_profiling.argtypes = [H, i]
    L.csa_reset_profile.argtypes = [H]
    L.csa_get_profile.argtypes = [H, ctypes.POINTER(ctypes.c_double), i, ctypes.POINTER(ctypes.c_long)]
    L.csa_set_halves.argtypes = [H, i]
    L.csa_set_rec1_max_batch.argtypes = [H, i]
    L.csa_debug_stage.argtypes = [H, i, i, _F, _F, _F, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_set_small_gemm_rows.argtypes = [i]
    L.csa_set_gemm_split.argtypes = [i]
    L.csa_stage_name.argtypes = [i]
    L.csa_stage_name.restype = ctypes.c_char_p
    Fp = ctypes.POINTER(ctypes.c_float)
    f = ctypes.c_float
    L.csa_train_create.argtypes = [ctypes.POINTER(CsaConfig), ctypes.POINTER(CsaParams), Fp, Fp, i, i, ctypes.POINTER(H)]
    L.csa_train_destroy.argtypes = [H]
    L.csa_train_num_params.argtypes = [H]
    L.csa_train_num_tensors.argtypes = [H]
    L.csa_train_param_info.argtypes = [H, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i), ctypes.POINTER(i), ctypes.POINTER(i)]
    L.csa_train_params.argtypes = [H]
    L.csa_train_params.restype = ctypes.c_void_p
    L.csa_train_sync_params.argtypes = [H, ctypes.c_void_p]
    L.csa_train_forward.argtypes = [H, i, i, _F, _F, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_train_forward_noise.argtypes = [H, i, i] + [_F] * 9 + [ctypes.c_void_p]
    L.csa_train_backward.argtypes = [H, i, i, _F, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_train_copy_state.argtypes = [H, i, i, _F, ctypes.c_void_p]
    L.csa_train_set_deferred.argtypes = [H, i]
    L.csa_train_flush_wgrad.argtypes = [H, _F, ctypes.c_void_p]
    L.csa_train_loss.argtypes = [H, i, i, f, f] + [_F] * 11 + [ctypes.c_void_p]
    L.csa_train_adam.argtypes = [H, _F, f, f, f, f, f, i, ctypes.c_void_p]
    L.csa_train_set_profiling.argtypes = [H, i]
    L.csa_train_reset_profile.argtypes = [H]
    L.csa_train_get_profile.argtypes = [H, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long), i]
    L.csa_train_stage_name.argtypes = [i]
    L.csa_train_stage_name.restype = ctypes.c_char_p
    PP = ctypes.POINTER(ctypes.POINTER(ctypes.c_float))
    L.csa_mlp_create.argtypes = [i, ctypes.POINTER(i), PP, PP, f, i, i, ctypes.POINTER(H)]
    L.csa_mlp_destroy.argtypes = [H]
    L.csa_mlp_forward.argtypes = [H, i, _F, _F, ctypes.c_void_p]
    L.csa_mlp_train_create.argtypes = [i, ctypes.POINTER(i), PP, PP, ctypes.c_float, i, i, ctypes.POINTER(H)]
    L.csa_mlp_train_destroy.argtypes = [H]
    L.csa_mlp_train_num_params.argtypes = [H]
    L.csa_mlp_train_num_params.restype = ctypes.c_long
    L.csa_mlp_train_copy_params.argtypes = [H, i, _F, ctypes.c_void_p]
    L.csa_mlp_train_forward.argtypes = [H, i, _F, _F, ctypes.c_void_p]
    L.csa_mlp_train_backward.argtypes = [H, _F, ctypes.c_float, _F, _F, ctypes.c_void_p]
    L.csa_mlp_train_adam.argtypes = [H, _F, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, i, ctypes.c_void_p]
    L.csa_cnn_create.argtypes = [i, i, i, i, i, i, PP, PP, i, ctypes.POINTER(H)]
    L.csa_cnn_destroy.argtypes = [H]
    L.csa_cnn_forward.argtypes = [H, i, _F, _F, ctypes.c_void_p]
    L.csa_cnn_reshape_to.argtypes = [i, i, i, i, _F, _F, ctypes.c_void_p]
    L.csa_cnn_reshape_from.argtypes = [i, i, i, i, _F, _F, ctypes.c_void_p]
    fl, PL, PI = ctypes.c_float, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_int)
    L.csa_cnn_train_create.argtypes = [i, i, i, i, i, i, PP, PP, i, fl, ctypes.POINTER(H)]
    L.csa_cnn_train_destroy.argtypes = [H]
    L.csa_cnn_train_num_params.argtypes = [H]
    L.csa_cnn_train_num_params.restype = ctypes.c_long
    L.csa_cnn_train_num_layers.argtypes = [H]
    L.csa_cnn_train_params.argtypes = [H]
    L.csa_cnn_train_params.restype = ctypes.c_void_p
    L.csa_cnn_train_get_params.argtypes = [H, _F, ctypes.c_void_p]
    L.csa_cnn_train_set_params.argtypes = [H, _F, ctypes.c_void_p]
    L.csa_cnn_train_get_act.argtypes = [H, i, i, _F, PI, ctypes.c_void_p]
    L.csa_cnn_train_layer_info.argtypes = [H, i, PL, PL, PI, PI, PI, PI, PI]
    L.csa_cnn_train_forward.argtypes = [H, i, _F, ctypes.c_void_p, _F, ctypes.c_void_p]
    L.csa_cnn_train_backward.argtypes = [H, _F, fl, _F, _F, ctypes.c_void_p]
    L.csa_cnn_train_adam.argtypes = [H, _F, fl, fl, fl, fl, i, ctypes.c_void_p]
    L.csa_gen_create.argtypes = [ctypes.POINTER(CsaGenConfig), ctypes.POINTER(CsaGenCoeffs), ctypes.POINTER(H)]
    L.csa_gen_destroy.argtypes = [H]
    L.csa_gen_dims.argtypes = [H, PI, PI, PI]
    L.csa_gen_batch.argtypes = [H, i] + [_F] * 11 + [ctypes.c_void_p]
    L.csa_crps.argtypes = [i, i, i, i, i, _F, _F, _F, _F, fl, fl, _F, _F, ctypes.c_void_p]
    L.csa_crps_backward.argtypes = [i, i, i, i, i, _F, _F, _F, _F, fl, fl, fl, _F, _F, ctypes.c_void_p]
    L.csa_spread_skill.argtypes = [i, i, i, i, i, _F, _F, _F, _F, ctypes.c_void_p, _F, ctypes.c_void_p]
    L.csa_eval_scratch_bytes.argtypes = [i, i, i, i]
    L.csa_eval_scratch_bytes.restype = ctypes.c_long
    L.csa_eval_metrics.argtypes = [i, i, i, _F, _F, i, ctypes.c_void_p, _F, ctypes.c_void_p]
    L.csa_eval_crps.argtypes = [i, i, i, i, _F, _F, i, ctypes.c_void_p, _F, ctypes.c_void_p]
    L.csa_derive_inputs.argtypes = [ctypes.c_long] + [_F] * 8 + [ctypes.c_void_p]
    U8 = ctypes.POINTER(ctypes.c_ubyte)
    L.csa_online_create.argtypes = [i, i, ctypes.POINTER(i), PP, PP, _F, _F, _F, U8, fl, fl, _F, U8, i, i, ctypes.POINTER(H)]
    L.csa_phys_create.argtypes = [i, i, i, i, i, i, PP, i, ctypes.POINTER(H)]
    L.csa_phys_rad_create.argtypes = [i, i, i, i, i, i, i, i, PP, i, ctypes.POINTER(H)]
    L.csa_phys_forward_noise.argtypes = [H, i, _F, _F, _F, _F, i, _F, _F, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_phys_debug_forward_srnn.argtypes = [H, i, _F, _F, _F, _F, i, _F, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_phys_debug_rnn3.argtypes = [H, i, i, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_phys_destroy.argtypes = [H]
    L.csa_phys_wrapped_create.argtypes = [i, i, i, PP, i, ctypes.POINTER(H)]
    L.csa_phys_wrapped_forward.argtypes = [H, i] + [_F] * 11 + [ctypes.c_void_p]
    L.csa_phys_forward.argtypes = [H, i, _F, _F, _F, _F, i, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_phys_tap.argtypes = [H, i, i, _F, ctypes.c_void_p]
    L.csa_phys_train_enable.argtypes = [H, i]
    L.csa_phys_train_num_params.argtypes = [H, ctypes.POINTER(i), ctypes.POINTER(i)]
    L.csa_phys_train_param_info.argtypes = [H, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i), ctypes.POINTER(i), ctypes.POINTER(i)]
    L.csa_phys_train_get_params.argtypes = [H, _F, ctypes.c_void_p]
    L.csa_phys_train_set_params.argtypes = [H, _F, ctypes.c_void_p]
    L.csa_phys_train_forward.argtypes = [H, i, i, _F, _F, _F, _F, i, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_phys_train_backward.argtypes = [H, i, i, _F, _F, _F, _F, i, _F, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_phys_train_adam_step.argtypes = [H, _F] + [ctypes.c_float] * 5 + [ctypes.c_void_p]
    L.csa_phys_train_loss.argtypes = [H, i, i, i, ctypes.c_float, ctypes.c_float] + [_F] * 11 + [ctypes.c_void_p]
    L.csa_phys_postprocess.argtypes = [H, i, _F, _F, _F, i, _F, _F, ctypes.c_void_p]
    L.csa_online_destroy.argtypes = [H]
    L.csa_online_dims.argtypes = [H, ctypes.POINTER(i), ctypes.POINTER(i)]
    L.csa_online_forward.argtypes = [H, i, _F, _F, ctypes.c_void_p]
    L.csa_stoch_gru5_create.argtypes = [i, i, Fp, Fp, Fp, Fp, Fp, i, ctypes.POINTER(H)]
    L.csa_stoch_lstm4_create.argtypes = [i, i, Fp, i, ctypes.POINTER(H)]
    L.csa_stoch_destroy.argtypes = [H]
    L.csa_stoch_gru5_forward.argtypes = [H, i, i, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_stoch_lstm4_forward.argtypes = [H, i, i, _F, _F, _F, _F, _F, _F, _F, ctypes.c_void_p]
    L.csa_stoch_enable_training.argtypes = [H]
    L.csa_stoch_num_params.argtypes = [H]
    L.csa_stoch_num_params.restype = ctypes.c_long
    L.csa_stoch_gru5_forward_train.argtypes = L.csa_stoch_gru5_forward.argtypes
    L.csa_stoch_lstm4_forward_train.argtypes = L.csa_stoch_lstm4_forward.argtypes
    L.csa_stoch_activation_floats.argtypes = [H, i, i]
    L.csa_stoch_activation_floats.restype = ctypes.c_long
    L.csa_stoch_set_activations.argtypes = [H, _F, i, i]
    L.csa_stoch_gru5_backward.argtypes = [H, i, i] + [_F] * 7 + [ctypes.c_void_p]
    L.csa_stoch_lstm4_backward.argtypes = [H, i, i] + [_F] * 10 + [ctypes.c_void_p]
    L.csa_last_error.restype = ctypes.c_char_p
    L.csa_version.restype = ctypes.c_char_p
    _lib = L
    return L


def last_error():
    return lib().csa_last_error().decode()
