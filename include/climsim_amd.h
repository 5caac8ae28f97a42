/*
 * climsim_amd.h -- C ABI of the MI355X-native per-column emulator path.
 *
 * This is the drop-in boundary for the reference's deployment wrappers
 * (peterukk/ClimSim, paths relative to the reference root):
 *
 *   csa_forward_packed   replaces  TorchScript NewModel.forward
 *                          stateless  rnn/save_wrapper.py:255-298      (x_main,x_sfc) -> (B,368)
 *                          stateful   rnn/save_wrapper_mem.py:499-545  (x_main,x_sfc,rnn1_mem) -> (B,368+nlev*nh_mem)
 *   csa_forward_tuple    replaces  model_wrapper.forward_base  rnn/utils.py:260-295
 *                          (x_main0,x_sfc0,rnn1_mem) -> (out_lev (B,nlev,6), out_sfc (B,8), rnn1_mem (nlev,B,nh_mem))
 *   csa_model_forward    replaces  RNN_autoreg.forward         rnn/models/models.py:432-608
 *                          [x_main_norm, x_sfc_norm, rnn_mem] -> (out (B,nlev,ny), out_sfc (B,8), rnn_mem)
 *
 * Conventions: fp32, contiguous, batch-first, raw physical units in
 * and out for the wrappers; inputs are never modified; ALL recurrent state (rnn1_mem) is owned
 * by the caller and passed / returned on every call; the handle holds weights, constants and
 * scratch only.  Every data pointer is a DEVICE pointer; `stream` is a hipStream_t (NULL =
 * default stream).  Calls are asynchronous on `stream`, allocate nothing and are safe to
 * capture into a hipGraph (single-stream path: csa_set_halves(h, 0)).  All functions return 0 on success or a negative csa_status; no
 * exception crosses the boundary.  One call at a time per handle.
 *
 * The legacy generation (the shipped rnn/v4_rnn*_wrapper*.pt artefacts) draws
 * hx2,cx2 = randn(B,nh) inside forward; here the draw is an explicit argument so that results
 * are reproducible and checkable (the Python facade draws it with torch when not given).
 */
#ifndef CLIMSIM_AMD_H
#define CLIMSIM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CSA_OK = 0,
    CSA_ERR_ARG = -1,          /* bad shape / null pointer / B out of range */
    CSA_ERR_UNSUPPORTED = -2,  /* configuration not implemented by the HIP path */
    CSA_ERR_HIP = -3,          /* HIP runtime error (see csa_last_error) */
    CSA_ERR_NOMEM = -4
} csa_status;

typedef struct {
    int32_t nlev, nx, nx_sfc, ny, ny_sfc;   /* 60, 15, 19, 5, 8 */
    int32_t nh1, nh2, nh_mem;               /* 128, 128, 16 (nh_mem 0 = stateless model) */
    int32_t use_lstm;        /* 1 LSTM, 0 GRU */
    int32_t legacy;          /* 1: shipped-artefact generation; 0: current RNN_autoreg */
    int32_t output_prune;    /* models.py:554-559 */
    int32_t mp_mode;         /* 1: liq/ice diagnosed from T; 0: none */
    int32_t snowhice_fix, qinput_prune, rh_prune, scrub_inf, scrub_out_nan;
    int32_t q_input_mode;    /* rnn/utils.py:262-272: 0 none; 1 include_q_input: specific humidity from (RH,T,p) is
                                appended as the LAST level input (nx counts it; raw x_main has nx-1 columns);
                                2 rh_to_q: it replaces RH (input 1) */
    int32_t add_stochastic_layer;/* models.py:405-412,464-474,521-534: rnn0 (down, noise init) -> rnn1 (up) -> stochastic
                                LSTM "rnn2" (MyStochasticLSTMLayer4, down).  LSTM, current generation, nh1 == nh2 */
    int32_t v5_input;        /* model_wrapper.preprocessing, rnn/utils.py:186-198: level input 2 = 1 - exp(-(qliq+qice) lbd_qn)
                                (qinput_prune zeroes qn of levels 0-14 BEFORE the transform), input 3 = the liquid fraction
                                hardtanh((T - 253.16) * 0.05, 0, 1) (models.py:260-266); needs csa_params.lbd_qn */
} csa_config;

/* HOST pointers, PyTorch state_dict layout (out_features,in_features); copied by csa_create. */
typedef struct {
    const float *xmean_lev, *xdiv_lev, *xmean_sca, *xdiv_sca, *lbd_qc, *lbd_qi;
    const float *yscale_lev, *yscale_sca, *hyam, *hybm;
    const float *mlp_initial_w, *mlp_initial_b;
    const float *mlp_surface1_w, *mlp_surface1_b, *mlp_surface2_w, *mlp_surface2_b;
    const float *mlp_toa1_w, *mlp_toa1_b, *mlp_toa2_w, *mlp_toa2_b;
    const float *rnn1_w_ih, *rnn1_w_hh, *rnn1_b_ih, *rnn1_b_hh;
    const float *rnn2_w_ih, *rnn2_w_hh, *rnn2_b_ih, *rnn2_b_hh;
    const float *mlp_latent_w, *mlp_latent_b, *mlp_output_w, *mlp_output_b;
    const float *mlp_surface_output_w, *mlp_surface_output_b;
    /* add_stochastic_layer only: rnn0 = nn.LSTM(nh1+nh_mem, nh1); rnn1_* then is nn.LSTM(nh1, nh2);
     * rnn2_weight_encoder (nh1+nh2, 5*nh2) in the reference's (in,out) layout; rnn2_w_* are unused */
    const float *rnn0_w_ih, *rnn0_w_hh, *rnn0_b_ih, *rnn0_b_hh;
    const float *rnn2_weight_encoder;
    const float *lbd_qn;     /* (nlev) v5_input only (models.py:171) */
} csa_params;

typedef struct csa_emulator csa_emulator;

/* Uploads/packs weights for the current HIP device and allocates scratch for up to
 * max_batch columns per call. */
int csa_create(const csa_config *cfg, const csa_params *host_params, int max_batch, csa_emulator **out);
int csa_destroy(csa_emulator *h);

/* Re-upload weights (training: after an optimiser step).  Host pointers, same layout. */
int csa_set_params(csa_emulator *h, const csa_params *host_params);

/* Width of one packed output row: 6*nlev + ny_sfc + nlev*nh_mem. */
int csa_packed_width(const csa_emulator *h);
int csa_max_batch(const csa_emulator *h);

/* Packed wrappers.  mem_in NULL iff nh_mem == 0.  hx2/cx2 (B,nh2) required iff legacy. */
int csa_forward_packed(csa_emulator *h, int B,
                       const float *x_main, const float *x_sfc, const float *mem_in,
                       const float *hx2, const float *cx2,
                       float *yout, void *stream);

/* Tuple wrapper of the current generation; mem_in/mem_out are (nlev,B,nh_mem). */
int csa_forward_tuple(csa_emulator *h, int B,
                      const float *x_main, const float *x_sfc, const float *mem_in,
                      float *out_lev, float *out_sfc, float *mem_out, void *stream);

/* Normalised-space model forward.  mem layout: legacy (B,nlev,nh_mem) sequence order,
 * current (nlev,B,nh_mem) level order. */
int csa_model_forward(csa_emulator *h, int B,
                      const float *x_main_n, const float *x_sfc_n, const float *mem_in,
                      const float *hx2, const float *cx2,
                      float *out, float *out_sfc, float *mem_out, void *stream);

/* RNN_autoreg.postprocessing(out, out_sfc, x_denorm) (models.py:273-339) on its own: normalised model outputs (B,nlev,ny),
 * (B,ny_sfc) and the raw level inputs x_denorm (B,nlev,nxd) (T first, qliq / qice at 2, 3; mp_mode -2 reads the LAST column
 * as the old specific humidity) -> out_lev (B,nlev,6) = [dT,dqv,dqliq,dqice,du,dv], out_sfc_denorm (B,ny_sfc).  No NaN scrub
 * (that is the wrapper's, rnn/utils.py:286).  mp_mode 0 is the identity upstream and is rejected here. */
int csa_postprocess(csa_emulator *h, int B, const float *out, const float *out_sfc, const float *x_denorm, int nxd,
                    float *out_lev, float *out_sfc_denorm, void *stream);

/* Stochastic variant (cfg.add_stochastic_layer): same as csa_forward_tuple / csa_model_forward with the three
 * N(0,1) draws the reference makes inside forward passed explicitly, in its draw order (models.py:466-468 and
 * models_torch_kernels.py:1497): hx0, cx0 (B,nh1) initial state of rnn0, eps (nlev,B,nh2). */
int csa_forward_tuple_noise(csa_emulator *h, int B, const float *x_main, const float *x_sfc, const float *mem_in,
                            const float *hx0, const float *cx0, const float *eps,
                            float *out_lev, float *out_sfc, float *mem_out, void *stream);
int csa_model_forward_noise(csa_emulator *h, int B, const float *x_main_n, const float *x_sfc_n, const float *mem_in,
                            const float *hx0, const float *cx0, const float *eps,
                            float *out, float *out_sfc, float *mem_out, void *stream);

/* Stateful + AR-noise packed wrapper (rnn/save_wrapper_mem.py:640-727): forward(x_main, x_sfc, rnn1_mem (B,nlev,nh_mem),
 * eps_prev (B,nlev,nh2)) -> yout (B, 6*nlev + ny_sfc + nlev*nh_mem + nlev*nh2) = [packed tendencies | surface | new memory of
 * the column | eps of the column], NaN -> 0.  The wrapper hands eps_prev to the model as the noise of its stochastic layer and
 * packs the eps the model returns; the AR(1) model classes that updated eps are commented out upstream
 * (models_torch_kernels.py:1251-1444), so the eps block returned here is the eps that was used (pass-through).  Model: the
 * add_stochastic_layer generation; hx0, cx0 (B,nh1) as in csa_forward_tuple_noise.  All state batch-first and caller-owned. */
int csa_forward_packed_noise(csa_emulator *h, int B, const float *x_main, const float *x_sfc, const float *mem_in,
                             const float *hx0, const float *cx0, const float *eps_prev, float *yout, void *stream);

/* Debug taps of the last call: rnn1 / rnn2 hidden sequences, (nlev,B,nh) level order. */
const float *csa_tap_rnn1(const csa_emulator *h);
const float *csa_tap_rnn2(const csa_emulator *h);

/* Optional per-kernel timing with HIP events on the call's own stream (bench.py's roofline
 * accounting).  While enabled, each forward records 7 events around its 6 launches and the NEXT
 * call (or csa_get_profile) collects them, which synchronises the host with the previous call:
 * measurement mode only.  Stages: prep, proj_gemm_rnn1, rec_rnn1, proj_gemm_rnn2, rec_rnn2, head. */
int csa_set_profiling(csa_emulator *h, int enable);
int csa_reset_profile(csa_emulator *h);
int csa_get_profile(csa_emulator *h, double *avg_ms /* [6] */, int n, long *calls);
const char *csa_stage_name(int i);

/* Two column halves on two streams with one fork / one join event (B >= 64; not the stochastic variant); results are
 * bit-identical to the single-stream path.  enable: 0 off, 1 on, 2 automatic (default: on from 640 columns, where it
 * measures 4-9 % faster; slower below).  Returns the new state. */
int csa_set_halves(csa_emulator *h, int enable);
/* Process-wide: projections with at most `rows` rows (= nlev * B) use the small-M split-K GEMM (32x32 tiles; default
 * 11,520 = 192 columns, the measured crossover; also settable with CSA_SMALL_GEMM_ROWS before the first call). */
int csa_set_small_gemm_rows(int rows);
/* Opt-in: the plain input projections (launch_proj_gemm above the small-GEMM threshold) run with every fp32 operand split
 * exactly into three bf16 values and six partial products per term on the bf16 matrix pipe, fp32 accumulation (gemm.hip;
 * the dropped cross terms are <= 2^-24 |a||b|, one fp32 rounding).  Default off (environment: CSA_GEMM_SPLIT_BF16=1): the
 * default is the fp32 MFMA chain. */
int csa_set_gemm_split(int on);
/* Largest batch that runs the recurrence with one column per workgroup (latency variant, LSTM nh <= 128; default 256 =
 * one column per CU).  0 forces the two-column kernel everywhere. */
int csa_set_rec1_max_batch(csa_emulator *h, int max_batch);

/* Stage-wise execution for parity evidence: runs ONE of the six launches of a forward call (0 prep, 1 projection rnn1,
 * 2 recurrence rnn1, 3 projection rnn2, 4 recurrence rnn2, 5 head + packing, 6 / 7 single cell step of rnn1 / rnn2) on caller-provided DEVICE inputs in the
 * library's internal layouts, so that each kernel can be teacher-forced with the reference's input of that stage:
 *   0: in0 x_main, in1 x_sfc, in2 mem_in (nullable), in3 / in4 hx2, cx2 (legacy) -> out0 X1 (nlev,B,nh1+nh_mem) sequence
 *      order (t = 0 = surface), out1 (4,B,max(nh1,nh2)) initial states [h0,c0 of rnn1 | h0,c0 of rnn2]
 *   1 / 3: in0 X (nlev*B,K) -> out0 pre-activations (nlev*B,4*nh), unit-major [i,g~,f,o] (GRU [r,z,n,0]), b_ih+b_hh folded
 *   2 / 4: in0 pre-activations, in1 h0, in2 c0 -> out0 hidden sequence (nlev,B,nh) in LEVEL order
 *   5: in0 rnn2 hidden sequence, in1 x_main, in2 x_sfc -> out0 packed output (B, csa_packed_width)
 *   6 / 7: in0 pre-activations (nlev*B,4*nh), in1 h_{t-1}, in2 c_{t-1} (nlev*B,nh) -> out0 h_t (nlev*B,nh): ONE cell step of
 *      rnn1 / rnn2 per row (the recurrent kernel's arithmetic without the chain of nlev dependent steps) */
int csa_debug_stage(csa_emulator *h, int stage, int B, const float *in0, const float *in1, const float *in2,
                    const float *in3, const float *in4, float *out0, float *out1, void *stream);

/* ---- training step (e) ---------------------------------------------
 * Replaces, for the current-generation LSTM with memory (mp_mode 1):
 *   model(inp_list) with autograd graph      rnn/utils.py:1098-1137   -> csa_train_forward (slot = step of the window)
 *   lossf + energy/water metrics + backward  rnn/utils.py:1203-1371, rnn/metrics.py:142-315
 *                                                                     -> csa_train_loss, csa_train_backward
 *   optim.step() (torch.optim.Adam)          rnn/utils.py:1377        -> csa_train_adam
 * Gradients accumulate into ONE caller-provided flat fp32 buffer in state_dict layout
 * (csa_train_param_info), which is also the buffer the single RCCL all-reduce of a data-parallel
 * step runs on.  The caller chains d_mem_in of slot tau into d_mem_out of slot tau-1 (TBPTT) and
 * zeroes `grads` at the start of a window.  hyai/hybi (nlev+1) are the hybrid interface coefficients
 * used by the energy / water closures. */
typedef struct csa_trainer csa_trainer;
int csa_train_create(const csa_config *cfg, const csa_params *host_params, const float *hyai, const float *hybi,
                     int max_batch, int max_window, csa_trainer **out);
int csa_train_destroy(csa_trainer *h);
int csa_train_num_params(const csa_trainer *h);
int csa_train_num_tensors(const csa_trainer *h);
int csa_train_param_info(const csa_trainer *h, int i, const char **name, int *offset, int *rows, int *cols);
float *csa_train_params(csa_trainer *h);                     /* device pointer, canonical flat parameters */
int csa_train_sync_params(csa_trainer *h, void *stream);     /* re-pack after writing csa_train_params() */
/* checkpoint / resume (rnn/train_rnn_rollout_torchscript_hydra.py:761-794,1003-1009): which = 0 parameters, 1 / 2 the two
 * Adam moments; dir = 0 copy out to buf, 1 copy in (device buffers of csa_train_num_params floats, state_dict order) */
int csa_train_copy_state(csa_trainer *h, int which, int dir, float *buf, void *stream);
int csa_train_forward(csa_trainer *h, int slot, int B, const float *x_main_n, const float *x_sfc_n,
                      const float *mem_in, float *out, float *out_sfc, float *mem_out, void *stream);
/* The stochastic variant (cfg.add_stochastic_layer; models.py:464-474,521-534: rnn0 down with noise initial state -> rnn1 up ->
 * MyStochasticLSTMLayer4 down): forward with the three N(0,1) draws of the reference passed explicitly, as in
 * csa_model_forward_noise: hx0, cx0 (B,nh1), eps (nlev,B,nh2).  `eps` is read again by csa_train_backward of the same slot:
 * the caller keeps it alive until then.  csa_train_backward / _adam / _copy_state work unchanged (the flat layout carries
 * rnn0.*, rnn1.*, rnn2.weight_encoder in state_dict order); the deferred weight-gradient mode is not available.  Ensemble
 * training (rnn/utils.py:1065-1075,1213) = B * E columns per call (inputs replicated member-major), d_out from csa_crps_backward. */
int csa_train_forward_noise(csa_trainer *h, int slot, int B, const float *x_main_n, const float *x_sfc_n, const float *mem_in,
                            const float *hx0, const float *cx0, const float *eps, float *out, float *out_sfc, float *mem_out,
                            void *stream);
int csa_train_backward(csa_trainer *h, int slot, int B, const float *d_out, const float *d_out_sfc,
                       const float *d_mem_out /* nullable */, float *d_mem_in /* nullable */, float *grads, void *stream);
/* Deferred weight gradients: with enable = 1 the backward calls skip the W_ih / W_hh gradient GEMMs and
 * csa_train_flush_wgrad does each of them ONCE over all (at most 8) time steps backpropagated since the last flush
 * (one contraction over T_w*nlev*B rows instead of T_w separate ones).  Call flush before using `grads`. */
int csa_train_set_deferred(csa_trainer *h, int enable);
int csa_train_flush_wgrad(csa_trainer *h, float *grads, void *stream);
/* pred/tgt (Tw*B,nlev,ny) normalised, *_sfc (Tw*B,ny_sfc); yto (Tw*B,nlev,6), yto_sfc physical targets;
 * x_raw (Tw*B,nlev,nx) raw inputs; x_sfc_n (Tw*B,nx_sfc) normalised.  scalars (device, 7 floats):
 * loss, huber, mse, mae, energy, water, precip_sum_mse.  d_pred/d_pred_sfc nullable (evaluation). */
int csa_train_loss(csa_trainer *h, int B, int Tw, float w_energy, float w_water,
                   const float *pred, const float *pred_sfc, const float *tgt, const float *tgt_sfc,
                   const float *yto, const float *yto_sfc, const float *x_raw, const float *x_sfc_n,
                   float *scalars, float *d_pred, float *d_pred_sfc, void *stream);
int csa_train_adam(csa_trainer *h, const float *grads, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, void *stream);
/* Optional per-stage timing of the training step with HIP events recorded on the call's own stream (bench.py's roofline of
 * the BPTT recurrence).  Stages (csa_train_stage_name): 0 fwd_rec (one launch per RNN per forward), 1 bwd_rec (one BPTT
 * launch per RNN per backward), 2 fwd_proj_gemm, 3 dx_gemm, 4 wgrad_flush (the four deferred weight-gradient contractions).
 * csa_train_get_profile synchronises with the recorded events: measurement mode only. */
#define CSA_TRAIN_NSTAGE 5
int csa_train_set_profiling(csa_trainer *h, int enable);
int csa_train_reset_profile(csa_trainer *h);
int csa_train_get_profile(csa_trainer *h, double *avg_ms /* [CSA_TRAIN_NSTAGE] */, long *launches /* nullable */, int n);
const char *csa_train_stage_name(int i);

/* ---- offline MLP baseline ------------------------------------------------------
 * baseline_models/MLP/training/HPO/baseline_v1/step2_retrain/step2_retrain.py:93-121: Dense stack with
 * LeakyReLU(alpha) and a split Dense(n_lin, linear) || Dense(rest, relu) output.  weights[l] is (dims[l+1], dims[l])
 * row-major (the transpose of the Keras kernel), HOST pointers; the two output Dense layers are stacked into one
 * (n_lin + n_relu, dims[nlayers-1]) matrix. */
typedef struct csa_mlp csa_mlp;
int csa_mlp_create(int nlayers, const int *dims, const float *const *weights, const float *const *biases,
                   float leaky_alpha, int n_lin_out, int max_batch, csa_mlp **out);
int csa_mlp_destroy(csa_mlp *h);
int csa_mlp_forward(csa_mlp *h, int B, const float *x, float *y, void *stream);

/* MLP baseline, one training step (step2_retrain.py:93-155: LeakyReLU stack, split head, loss 'mse', keras Adam).  Flat
 * parameters / gradients in layer order [W_l (out,in) | b_l]; grad_scale multiplies dLoss/dy (data-parallel shares). */
typedef struct csa_mlp_trainer csa_mlp_trainer;
int csa_mlp_train_create(int nlayers, const int *dims, const float *const *weights, const float *const *biases,
                         float leaky_alpha, int n_lin_out, int max_batch, csa_mlp_trainer **out);
int csa_mlp_train_destroy(csa_mlp_trainer *h);
long csa_mlp_train_num_params(const csa_mlp_trainer *h);
int csa_mlp_train_copy_params(csa_mlp_trainer *h, int dir /* 0 out, 1 in */, float *buf, void *stream);
int csa_mlp_train_forward(csa_mlp_trainer *h, int B, const float *x, float *y, void *stream);
int csa_mlp_train_backward(csa_mlp_trainer *h, const float *y_true, float grad_scale, float *loss_out, float *grads, void *stream);
int csa_mlp_train_adam(csa_mlp_trainer *h, const float *grads, float lr, float beta1, float beta2, float eps, int step,
                       void *stream);

/* ---- offline CNN baseline, forward --------------------------------------------------
 * baseline_models/CNN/training/hpo_train.py:124-200: 12 residual blocks of two Conv1D(406,3,same)+ReLU plus a 1x1
 * projection of the block input, Conv1D(10,1,elu), Dense(2,linear) || Dense(8,relu).  weights/biases: HOST
 * pointers in PyTorch Conv1d layout (cout,cin,k), 3 per block (conv_a, conv_b, residual), then the pre-output
 * 1x1 conv, then the two Dense layers stacked as one (cout,cout) matrix. */
typedef struct csa_cnn csa_cnn;
int csa_cnn_create(int depth, int nlev, int cin, int width, int cout, int n_lin, const float *const *weights,
                   const float *const *biases, int max_batch, csa_cnn **out);
int csa_cnn_destroy(csa_cnn *h);
int csa_cnn_forward(csa_cnn *h, int B, const float *x, float *y, void *stream);

/* data-format adapters either side of the CNN (climsim_utils/data_utils.py:2104-2175): flat (N, nprof*nlev + nscal)
 * <-> channels-last (N, nlev, nprof + nscal); scalars are repeated over the levels one way and level-averaged back */
int csa_cnn_reshape_to(int N, int nlev, int nprof, int nscal, const float *flat, float *chan, void *stream);
int csa_cnn_reshape_from(int N, int nlev, int nprof, int nscal, const float *chan, float *flat, void *stream);

/* ---- CNN baseline, one training step (BASELINE.json configs[3]) ------------------------------------------------
 * baseline_models/CNN/training/hpo_train.py:124-236: forward with Dropout(p) after both activations of a block,
 * loss mae_adjusted (:118-120), keras Adam.  Parameters and gradients are flat fp32 device buffers in the GEMM layout
 * (per layer: W (cout_p, k*cin_p) with channels padded to multiples of 8 / 4, then b (cout_p)); csa_cnn_train_layer_info
 * gives the offsets.  masks: uint8 keep-flags (2*depth, B*nlev, width), drawn by the caller (NULL: no dropout).
 * grad_scale multiplies dLoss/dy (data-parallel shares).  loss_out: one device float (nullable). */
typedef struct csa_cnn_trainer csa_cnn_trainer;
int csa_cnn_train_create(int depth, int nlev, int cin, int width, int cout, int n_lin, const float *const *weights,
                         const float *const *biases, int max_batch, float dropout, csa_cnn_trainer **out);
int csa_cnn_train_destroy(csa_cnn_trainer *h);
long csa_cnn_train_num_params(const csa_cnn_trainer *h);
int csa_cnn_train_num_layers(const csa_cnn_trainer *h);
float *csa_cnn_train_params(csa_cnn_trainer *h);
int csa_cnn_train_get_params(csa_cnn_trainer *h, float *dst, void *stream);       /* device-to-device copies of the */
int csa_cnn_train_set_params(csa_cnn_trainer *h, const float *src, void *stream); /* flat parameter buffer */
/* debug tap: saved activation of the last forward; which 0 = after conv_a+ReLU+dropout, 1 = after conv_b+..., 2 = block output */
int csa_cnn_train_get_act(csa_cnn_trainer *h, int block, int which, float *dst, int *ld, void *stream);
int csa_cnn_train_layer_info(const csa_cnn_trainer *h, int i, long *w_off, long *b_off, int *cout, int *cin, int *k,
                             int *cout_p, int *cin_p);
int csa_cnn_train_forward(csa_cnn_trainer *h, int B, const float *x, const unsigned char *masks, float *y_out, void *stream);
int csa_cnn_train_backward(csa_cnn_trainer *h, const float *y_true, float grad_scale, float *loss_out, float *grads,
                           void *stream);
int csa_cnn_train_adam(csa_cnn_trainer *h, const float *grads, float lr, float beta1, float beta2, float eps, int step,
                       void *stream);

/* ---- training-data pipeline (the training twin of row a1 + target construction) -------------
 * generator_xy.__getitem__ rnn/utils.py:2238-2371 as one device pass over a loaded chunk of N = ntime*ncol samples.
 * The HDF5 reading stays with the host; arrays are device pointers. */
typedef struct {
    int32_t nlev, nx_in, nx_sfc_in, ny_sfc;         /* file layout: input_lev (.,nlev,nx_in), input_sca (.,nx_sfc_in), output_lev (.,nlev,6) */
    int32_t remove_past_sfc_inputs;                 /* np.delete(x_sfc, (17..21)) :2172-2173 */
    int32_t snowhice_fix, rh_prune, qinput_prune, output_prune;
    int32_t q_mode;                                 /* 0 none, 1 include_q_input (append), 2 rh_input_to_q (replace RH) :2183-2194 */
    int32_t cld_inp_transformation;                 /* 0 none, 1 exp, 2 sqrt */
    int32_t v4_to_v5_inputs;                        /* qn + liquid fraction instead of qliq, qice :2211-2231 */
    int32_t apply_new_input_scaling, reverse_input_norm, reverse_output_norm;
    int32_t mp_mode;                                /* targets: 0 six outputs, 1 qn (5 outputs), -1 qn + liq_frac, -2 total water */
} csa_gen_config;
typedef struct {                                    /* HOST pointers; unused ones may be NULL */
    const float *xmean_lev, *xdiv_lev;              /* xcoeffs (nlev, nx_out) */
    const float *xmean_sca, *xdiv_sca;              /* (nx_sfc_out) */
    const float *yscale_lev, *yscale_sca;           /* ycoeffs (nlev, ny_out), (ny_sfc) */
    const float *lbd_qc, *lbd_qi, *lbd_qn, *hyam, *hybm;   /* (nlev) */
    const float *xref_mean, *xref_div, *xsref_mean, *xsref_div, *yref_lev, *yref_sca;   /* xcoeffs_ref / ycoeffs_ref */
} csa_gen_coeffs;
typedef struct csa_generator csa_generator;
int csa_gen_create(const csa_gen_config *cfg, const csa_gen_coeffs *coeffs, csa_generator **out);
int csa_gen_destroy(csa_generator *h);
int csa_gen_dims(const csa_generator *h, int *nx_out, int *nx_sfc_out, int *ny_out);
/* -> (x_lev, x_sfc, y_lev, y_sfc, x_lev_denorm, y_lev_denorm, y_sfc_denorm) of __getitem__, all device buffers */
int csa_gen_batch(csa_generator *h, int N, const float *x_lev, const float *x_sfc, const float *y_lev, const float *y_sfc,
                  float *x_lev_n, float *x_sfc_n, float *y_lev_n, float *y_sfc_n, float *x_lev_denorm,
                  float *y_lev_denorm, float *y_sfc_denorm, void *stream);

/* ---- ensemble score, evaluation --------------------------------------------------------------
 * rnn/metrics.py:535-626 CRPS(y, y_sfc, y_pred, y_sfc_pred, timesteps, beta, alpha): energy-score form over the
 * concatenated [level | surface] outputs.  y (T*B, D_lev), y_sfc (T*B, D_sfc): truth; y_pred (T*E*B, D_lev), y_sfc_pred:
 * ensemble outputs ordered (time, member, column).  scratch: 2*T*B device floats; out: 3 device floats
 * [CRPS, skill term, spread term]. */
int csa_crps(int T, int B, int E, int D_lev, int D_sfc, const float *y, const float *y_sfc, const float *y_pred,
             const float *y_sfc_pred, float beta, float alpha, float *scratch, float *out, void *stream);

/* Gradient of csa_crps w.r.t. the ensemble outputs (the score as training loss, rnn/utils.py:1213; upstream: autograd through
 * torch.cdist): d_y_pred (T*E*B, D_lev), d_y_sfc_pred (T*E*B, D_sfc) are WRITTEN; gscale = dLoss/dCRPS; zero distances
 * contribute zero, as torch's cdist backward does. */
int csa_crps_backward(int T, int B, int E, int D_lev, int D_sfc, const float *y, const float *y_sfc, const float *y_pred,
                      const float *y_sfc_pred, float beta, float alpha, float gscale, float *d_y_pred, float *d_y_sfc_pred,
                      void *stream);

/* rnn/metrics.py:509-533 compute_spread_skill_ratio and rnn/metrics.py:628-699 CRPS_l1 on the same tensors as csa_crps.
 * scratch: 32 KB device; out: 4 device floats [spread (with the sqrt((E+1)/E) correction), RMSE of the member mean,
 * CRPS_l1 = mean|z_e - z| - 0.5 mean|z_0 - z_1|, its skill term]. */
int csa_spread_skill(int T, int B, int E, int D_lev, int D_sfc, const float *y, const float *y_sfc, const float *y_pred,
                     const float *y_sfc_pred, void *scratch, float *out, void *stream);

/* ---- evaluation scores of data_utils ---------------------------------------------------------
 * climsim_utils/data_utils.py:1843-1935 calc_MAE / calc_RMSE / calc_R2 / calc_bias / calc_CRPS.  pred, target:
 * (T, G, L) device floats (time, grid column, level; scalars L = 1); reductions over time per (grid, level) cell, then
 * with avg_grid != 0 the mean over grid.  csa_eval_metrics: out (4, G, L) or (4, L), rows MAE, RMSE, R2, bias.
 * csa_eval_crps: samplepreds (T, G, L, S), out (G, L) or (L).  scratch: csa_eval_scratch_bytes(T, G, L, S) device bytes
 * (S = 0 for csa_eval_metrics). */
long csa_eval_scratch_bytes(int T, int G, int L, int S);
int csa_eval_metrics(int T, int G, int L, const float *pred, const float *target, int avg_grid, void *scratch,
                     float *out, void *stream);
int csa_eval_crps(int T, int G, int L, int S, const float *samplepreds, const float *target, int avg_grid,
                  void *scratch, float *out, void *stream);

/* ---- derived loader inputs (climsim_utils/data_utils.py:654-707, get_xrdata) ------------------------------------------------------
 * Variables a variable set names but the files do not store, computed from the stored state on read: state_rh = q / qvs(T, p)
 * with the float64 saturation polynomials eliq / eice (:19-43, :662-673), liq_partition = clip((T - 253.16) / 20, 0, 1)
 * (:684-690), state_qn = q0002 + q0003 (:692-707; the *_prvphy pairs likewise).  n cells, device pointers; every output is
 * optional (NULL = not wanted) and only the inputs its formula reads are required. */
int csa_derive_inputs(long n, const float *state_t, const float *state_q0001, const float *state_pmid, const float *q2,
                      const float *q3, float *state_rh, float *liq_partition, float *state_qn, void *stream);

/* ---- generic online wrapper (SURVEY section 8b "generic online": forward(x (B, n_in)) -> (B, 368)) ------------------
 * online_testing/model_postprocessing/v4_nn_wrapper.ipynb cell 5 (NewModel.preprocessing / forward / postprocessing)
 * around online_testing/baseline_models/MLP_v2rh/training/mlp.py:25-67 (Linear + ReLU layers, final Linear, ReLU on the
 * last n_relu_tail outputs); host contract online_testing/README.md:47-50.
 * Per input column j: x = in_lbd[j] != 0 ? 1 - exp(-x in_lbd[j]) : x;  x = (x - in_sub[j]) / in_div[j];  NaN/Inf -> 0;
 * in_flags[j] bit 0: x = 0 (pruned level), bit 1: x = clamp(x, clip_lo, clip_hi).  Per output column j:
 * y = (out_keep[j] ? y : 0) / out_scale[j].  weights[l]: (dims[l+1], dims[l]) row-major HOST pointers, dims[0] = n_in
 * (any value; padded to a multiple of 4 internally), dims[1..nlayers-1] multiples of 4.  x is not modified. */
typedef struct csa_online csa_online;
int csa_online_create(int n_in, int nlayers, const int *dims, const float *const *weights, const float *const *biases,
                      const float *in_sub, const float *in_div, const float *in_lbd, const unsigned char *in_flags,
                      float clip_lo, float clip_hi, const float *out_scale, const unsigned char *out_keep,
                      int n_relu_tail, int max_batch, csa_online **out);
int csa_online_destroy(csa_online *h);
int csa_online_dims(const csa_online *h, int *n_in, int *n_out);
int csa_online_forward(csa_online *h, int B, const float *x, float *y, void *stream);

/* ---- physRNN "Hidden" model, forward -----------------------------------------------------------
 * rnn/models/models_phys.py:1586-1823 (forward), :414-748 (microphysics_decode), rnn/layers.py:117-168 (pressure layers),
 * in the geometry of the shipped rnn/saved_models/physRNN-Hidden_*_script_cpu.pt artefacts (GRU 128/128, nx = 21,
 * ilev_crm = 10, mp_ncol = 16, 15 + 1 memory channels on 50 levels).  w: HOST pointers, PyTorch layouts, in this order:
 *   hyam(60) hybm(60) hyai(61) hybi(61) yscale_lev(60,5) yscale_sca(8) xdiv_sca(nx_sfc) xmean_sca(nx_sfc)
 *   mlp_initial.{weight (nh,nx+1), bias} mlp_surface1.{weight (nh,nx_sfc), bias}
 *   rnn1.{weight_ih_l0 (3nh,nh+15), weight_hh_l0, bias_ih_l0, bias_hh_l0} rnn2.{weight_ih_l0 (3nh,nh), weight_hh_l0, bias_ih_l0, bias_hh_l0}
 *   mlp_latent.{w (15,nh), b} mlp_output.{w (5,15), b} mlp_surface_output_rad.{w (6,nh), b} mlp_output_rad.{w (1,nh), b}
 *   mlp_precip_release.{w (1,nh), b}
 *   then {weight (16,nh), bias} of mlp_qv_crm, mlp_qn_crm, mlp_t_crm, mlp_subgrid_area_frac, mlp_massflux, mlp_eddy_diff,
 *   mlp_qice_crm, mlp_sed_qn_crm, mlp_evap_prec_crm, mlp_evap_cond_vapor_crm, mlp_mp_aa_crm        (52 pointers)
 * Forward: device pointers; x_main (B,60,nx) and x_sfc (B,nx_sfc) normalised, rnn_mem (B,50,16) (last channel = stored
 * water), x_denorm (B,60,nxd) raw inputs (T first, qliq, qice at 2, 3, qv last), hx2 (B,nh) = the N(0,1) draw the
 * reference makes inside forward for rnn2's initial state.  Outputs out_lev (B,60,5), out_sfc (B,8), mem_out (B,50,16). */
typedef struct csa_phys csa_phys;
int csa_phys_create(int nx, int nx_sfc, int nh, int ilev_crm, int mp_ncol, int nh_mem0, const float *const *w,
                    int max_batch, csa_phys **out);
int csa_phys_destroy(csa_phys *h);
int csa_phys_forward(csa_phys *h, int B, const float *x_main, const float *x_sfc, const float *rnn_mem,
                     const float *x_denorm, int nxd, const float *hx2, float *out_lev, float *out_sfc, float *mem_out,
                     void *stream);
int csa_phys_tap(csa_phys *h, int which, int B, float *dst, void *stream);
/* The module's exported `postprocessing(out, out_sfc, x_denorm)` (rnn/models/models.py:273-339, mp_mode 1): out (B,60,5)
 * normalised -> out6 (B,60,6) physical [dT, dqv, dqliq, dqice, du, dv] (cloud-water tendency split by the temperature ramp
 * at the updated temperature), out_sfc (B,8) / yscale_sca -> out_sfc_denorm. */
int csa_phys_postprocess(csa_phys *h, int B, const float *out, const float *out_sfc, const float *x_denorm, int nxd,
                         float *out6, float *out_sfc_denorm, void *stream);

/* The radiation graphs of the same model family (`use_physrad`: physRNN-Hidden_*_num4050 / num71535 / num83000 / num5730 /
 * num62104): 21 level inputs of which the first 18 and the layer pressure feed mlp_initial, both GRUs over the 50 CRM
 * levels, surface inputs aux[0:6] and aux[11:19], and instead of the two radiative Linear heads the serialised
 * `radiative_transfer` (LW gas-optics MLP rnn/layers.py gasopt_mlp + no-scattering solver rnn/models/physics_rad.py:96,
 * learned SW optical properties + two-stream :139 + adding :332, E3SM effective radii rnn/models/physics_rad_e3sm.py:13,:62).
 * flags: CSA_PHYS_MCICA          mp_ncol 4, every g-point samples a sub-column (physics_rad.py:533); without it mp_ncol 16
 *                                and g-point g sees sub-column g;
 *        CSA_PHYS_LIQ_FRAC_HEAD  cloud liquid fraction = sigmoid(mlp_liq_frac_crm(rnn2 output)) instead of the temperature ramp;
 *        CSA_PHYS_STOCHASTIC     rnn3 = MyStochasticGRULayer5(nh, nh) (rnn/models_torch_kernels.py:834-891) over rnn2's output:
 *                                the heads read rnn2_out * rnn3_out, the precipitation-release head reads rnn3's last state;
 *                                such a handle needs csa_phys_forward_noise.
 *        CSA_PHYS_PHYSRAD        the physRNN_physRad-* generation of the graph (first geometry: nreg 16, GRU 128/128, e.g.
 *                                physRNN_physRad-16_nreg16_*_num14751_BEST_script_cpu.pt): region 0 of the mp_ncol = nreg regions is
 *                                clear sky -- mlp_qn_crm and mlp_evap_cond_vapor_crm are (nreg-1, nh) --, no sub-grid temperature
 *                                (the mlp_t_crm pair is NULL, mlp_eddy_diff is (1, nh)), latent heating from area-summed rates,
 *                                vapour mixing ratio q/(1-q), un-squared solar weights, and rnn_mem / mem_out are LEVEL-MAJOR
 *                                (50, B, 16) as that graph's forward takes and returns them.  Needs LIQ_FRAC_HEAD; with MCICA (nreg 4) the g-points
 *                                sample the cloudy regions 1.. only (fractions renormalised).
 *        CSA_PHYS_LATER_EXPORT   a later revision of the serialised scheme (e.g. physRNN_physRad-16_nreg16_*neur112-112*_num34341):
 *                                the LW downward sweep gets its own layer source (the first exports feed it the upward one), and the
 *                                `xmax` slot of the pointer list carries the gas-optics input range (buffer `xdiv`) itself.
 *        CSA_PHYS_CLOUD_OPTICS_LW cloud LW optical depth per unit path = ReLU(cloud_optics_lw([(T-160)/180, r_ice/125, r_liq/13.5,
 *                                new memory])) instead of the liquid / ice rule (num88955); two more pointers {w (16,19), b} at the end.
 *        CSA_PHYS_SW_GAS         the physics_rad_e3sm generation (physRNN_physRad-16_nreg16_*neur128-128*_num94634; the geometry of
 *                                the frozen *_wrapped exports): no SW head MLP (its four pointer slots may be NULL) -- SW optical
 *                                depth from two gas-optics MLPs 7 -> 32 -> 32 -> 112 (absorption, Rayleigh), evaluated for the
 *                                humidity of the two largest regions of each level and averaged, reduced 112 -> 16; Slingo liquid
 *                                / Ebert-Curry ice cloud optics per region; the LW downward sweep reads the upward source (set
 *                                LATER_EXPORT for the `xdiv` slot only -- this flag wins for the source).  One more pointer at the
 *                                end: a block of 17648 floats = gas_optics_model_sw1.{xmin (7, padded to 8), xdiv (8)}, then for
 *                                sw1 and sw2 {mlp1.w (32 rows of 12: 7 used), b (32), mlp2.w (32 rows of 36: 32 used), b, mlp3.w
 *                                (128 rows of 36: 112 x 32 used), b (128), ystd (128), ymean (128)}, gas_optics_sw_reduce1 {w (16 rows
 *                                of 132: 112 used), b (16)}, _reduce2 likewise -- all padding zero --, then the cloud coefficients
 *                                (12,16): Slingo A..F and Ebert-Curry a..f spread over the 16 g-points
 *                                (climsim_amd/physrnn.py builds it; oracle/physrnn_rad_ref.py band_table has the band limits).
 *        CSA_PHYS_CLOUD_OPTICS_SW with SW_GAS and CLOUD_OPTICS_LW (num88741): SW cloud optics learned as well -- [extinction per unit
 *                                path (ReLU) | single-scattering albedo (sigmoid) | asymmetry (sigmoid)] x 16 g-points =
 *                                cloud_optics_sw2(cloud_optics_sw(x)) on the LW layer's 19 inputs; the two Linear layers have no
 *                                activation between them and travel COMPOSED: two more pointers {w (48,19) = W2 W1, b (48) =
 *                                W2 b1 + b2} after the SW_GAS block.  This graph keeps LATER_EXPORT's own LW downward source.
 * nx = 21 or 16 level inputs (the last three before q_v and the pressure feature bypass mlp_initial), GRU 128 / 112 / 96.
 * Same handle type: csa_phys_forward / _tap (50 levels) / _destroy apply; x_sfc is (B, naux = 19), x_denorm needs columns
 * 12..14 = O3, CH4, N2O.
 * w (HOST pointers): the first 24 of csa_phys_create's list (hyam ... mlp_output.b), mlp_precip_release.{w, b}, the 11
 * decoder heads {weight (mp_ncol,nh), bias}, then lbd_qn (60), yscale_sca_rad (6), sw_solar_weights (16),
 * gas_optics_model_lw.{xmin (18), xmax (18), ymean (128), ystd (128), mlp1.{w (64,18), b}, mlp2.{w (64,64), b},
 * mlp3.{w (256,64), b}}, gas_optics_lw_reduce1.{w (16,128), b}, gas_optics_lw_reduce2.{w, b},
 * mlp_sw_optprops1.{w (32,24), b}, mlp_sw_optprops2.{w (48,32), b}  (69 pointers), then with CSA_PHYS_LIQ_FRAC_HEAD
 * mlp_liq_frac_crm.{w (mp_ncol,nh), b}, then with CSA_PHYS_STOCHASTIC rnn3.{weight_ih (nh,3nh), weight_zh (nh,3nh),
 * weight_encoder (nh,2nh)} in the reference's (in, out) layout.
 * csa_phys_forward_noise: csa_phys_forward + hx1 (B,nh), rnn3's initial state, and eps3 (50,B,nh), its noise: the two further
 * N(0,1) draws the reference makes inside forward (both nullable for a handle without rnn3). */
enum { CSA_PHYS_MCICA = 1, CSA_PHYS_LIQ_FRAC_HEAD = 2, CSA_PHYS_STOCHASTIC = 4, CSA_PHYS_PHYSRAD = 8, CSA_PHYS_LATER_EXPORT = 16,
       CSA_PHYS_CLOUD_OPTICS_LW = 32, CSA_PHYS_SW_GAS = 64, CSA_PHYS_CLOUD_OPTICS_SW = 128, CSA_PHYS_RAD_UPDATED_QV = 256, CSA_PHYS_SW_HEAD = 512 };
int csa_phys_rad_create(int nx, int naux, int nh, int ilev_crm, int mp_ncol, int nh_mem0, int ng, int flags,
                        const float *const *w, int max_batch, csa_phys **out);
int csa_phys_forward_noise(csa_phys *h, int B, const float *x_main, const float *x_sfc, const float *rnn_mem,
                           const float *x_denorm, int nxd, const float *hx2, const float *hx1, const float *eps3,
                           float *out_lev, float *out_sfc, float *mem_out, void *stream);
/* Test hooks for graphs whose stochastic layer is chaotic on the test inputs (no two float32 implementations agree end to
 * end): the forward pass with rnn3's output srnn (50,B,nh) supplied (teacher forcing), and the handle's rnn3 alone on
 * caller-supplied x (T,B,nh), h0 (B,nh), eps (T,B,nh) -> out (T,B,nh), T*B <= 50*max_batch. */
int csa_phys_debug_forward_srnn(csa_phys *h, int B, const float *x_main, const float *x_sfc, const float *rnn_mem,
                                const float *x_denorm, int nxd, const float *hx2, const float *srnn,
                                float *out_lev, float *out_sfc, float *mem_out, void *stream);
int csa_phys_debug_rnn3(csa_phys *h, int T, int B, const float *x, const float *h0, const float *eps, float *out, void *stream);

/* ---- the FROZEN exports rnn/saved_models/physRNN_physRad-*_nx21_*_script_{cpu,gpu}_wrapped.pt (82 of the 114 shipped artefacts, the
 * modules an E3SM host loads through its libtorch binding): rnn/utils.py::model_wrapper (:72-295) inlined by torch.jit.freeze around
 * the "nx21" generation of rnn/models/models_phys.py::physical_RNN_autoreg (:1586-1823, decoder :414-748, optics :816-1270, solver
 * :1272-1490).  Replaces the export's forward, same signature, raw physical units in and out:
 *     forward(x_main0 (B,60,20), x_sfc0 (B,19), rnn1_mem (50,B,16)) -> (out_lev (B,60,6), out_sfc (B,8), rnn1_mem (50,B,16))
 * csa_phys_wrapped_create: nh = 128; ng = number of sub-grid regions = g-points of the export (12, 14 or 16); weights as HOST pointers
 * under the names the constants had before freezing (tests/golden/frozen_extract.py recovers them from the serialised code), with
 * every per-region / per-g-point axis zero-padded to 16 by the caller (padded regions: area-fraction bias -1e30; padded g-points:
 * solar weight 0, Planck-fraction bias -1e30):
 *   hyam (60), hybm (60), hyai (61), hybi (61), yscale_lev (60,5), yscale_sca (8), xdiv_sca (19), xmean_sca (19),
 *   mlp_initial.{w (nh,19), b}, mlp_surface1.{w (nh,14), b}, rnn1.{weight_ih_l0 (3nh, nh+15), weight_hh_l0, bias_ih_l0, bias_hh_l0},
 *   rnn2.{...}, mlp_latent.{w (15,nh), b}, mlp_output.{w (5,15), b}, mlp_precip_release.{w (1,nh), b},
 *   {mlp_qv_crm, mlp_qn_crm, mlp_t_crm, mlp_subgrid_area_frac, mlp_massflux, mlp_eddy_diff, mlp_qice_crm, mlp_sed_qn_crm,
 *    mlp_evap_prec_crm, mlp_evap_cond_vapor_crm, mlp_mp_aa_crm}.{w (16,nh), b (16)},
 *   yscale_sca_rad (6), solar weights (16, as folded into the export: they sum to 1), gas_optics_model_lw.{xmin (18), xdiv (18),
 *   ymean (128), ystd (128), mlp1.{w (64,18), b}, mlp2.{w (64,64), b}, mlp3.{w (256,64), b}}, gas_optics_lw_reduce1.{w (16,128), b},
 *   gas_optics_lw_reduce2.{w (16,128), b}, the SW gas-optics block (csrc/phys.h SWX_*: input range, two 7-32-32-16 models whose last
 *   layer may give ngk <= 16 k-points, then gas_optics_sw_reduce1 / 2 as (16,16) + (16), zero when the export has no reduction), the
 *   Slingo / Ebert-Curry table per g-point (12,16) -- or, when misc[6] = 1, per BAND (12,4) followed by the learned band -> g-point
 *   matrix (4,16) --, misc (8 floats) [n_ir, n_mix_end, mix_near, mix_vis: the split of the g-points into near-infrared | mixed |
 *   visible and the weights of the mixed ones; ngk: k-points behind the SW coin that the reduction maps to the g-points, 0 = none;
 *   1 = the ice SW optics read the ice effective radius (0: the liquid one, as first serialised); 1 = band matrix; bit mask: 1 the
 *   first surface output is the downward (not net) shortwave, 2 cloud LW optics on the liquid fraction of the updated sub-grid
 *   temperature, 4 the SW head sees the updated cloud water, 8 decoder without a sub-grid temperature (the physRad decoder: zero
 *   mlp_t_crm head, the one eddy-diffusivity row repeated), 16 region 0 holds no condensate, 32 cloud water paths of the radiation
 *   scheme from the sub-grid cloud water before the step, 64 radiation on the temperature before the step, 128 the release / surface
 *   heads read rnn2's last state times the third RNN's (num36398)].  With CSA_PHYS_SW_GAS (num27378, num45826, num74834) the SW block is the
 *   unfrozen generation's (csa_phys_rad_create's CSA_PHYS_SW_GAS block: 112 k-points, mean of the two humidity variants).  With CSA_PHYS_SW_HEAD (earlier exports: num8701, num75599, num82174)
 *   the SW block and the cloud table are replaced by mlp_sw_optprops1.{w (32,24), b}, mlp_sw_optprops2.{w (48,32), b}, lbd_qn (60).  Then the
 *   wrapper's xmean_lev (60,21), xdiv_lev (60,21),
 *   lbd_qc (60), lbd_qi (60)   (71 pointers); with CSA_PHYS_LIQ_FRAC_HEAD mlp_liq_frac_crm.{w (16,nh), b}, with CSA_PHYS_STOCHASTIC
 *   rnn3.{weight_ih, weight_zh, weight_encoder}.  CSA_PHYS_RAD_UPDATED_QV: radiation reads the updated grid-mean q_v.
 * csa_phys_wrapped_forward: the random draws the export makes inside forward are ARGUMENTS (device pointers): hx2 (B,nh) rnn2's initial
 * state; hx1 (B,nh), eps3 (50,B,nh) the stochastic third RNN's state and noise (CSA_PHYS_STOCHASTIC, else null); mask_u (60,B,ngk or ng) the
 * uniform field of the SW humidity coin (`torch.rand_like(tau) < 0.5` picks the largest region's humidity; exports without the coin --
 * CSA_PHYS_SW_HEAD, CSA_PHYS_SW_GAS -- take null).  srnn: test hook, null in
 * production (the third RNN's output supplied: that layer is chaotic on synthetic inputs, see tests/test_physrnn_frozen.py). */
int csa_phys_wrapped_create(int nh, int ng, int flags, const float *const *w, int max_batch, csa_phys **out);
int csa_phys_wrapped_forward(csa_phys *h, int B, const float *x_main0, const float *x_sfc0, const float *rnn1_mem, const float *hx2,
                             const float *hx1, const float *eps3, const float *mask_u, const float *srnn, float *out_lev,
                             float *out_sfc, float *mem_out, void *stream);

/* ---- physRNN training (the trainable model of rnn/train_rnn_rollout_torchscript_hydra.py:553-554) ----------
 * Backward of the non-radiative Hidden graph (csa_phys_create handles): rnn/models_phys.py:226-412 (forward) and :414-748
 * (microphysics_decode), differentiated by hand; pinned by torch autograd through oracle/physrnn_ref.py (tests/test_physrnn_train.py).
 * csa_phys_train_enable  builds the training state from the weights given at create: a flat parameter vector in state_dict order
 *                        (csa_phys_train_param_info: name, offset, rows, cols of tensor i), Adam moments, the kernel layouts gathered
 *                        from it, and `nslots` sets of activation buffers for max_batch columns (one pending forward per slot: a
 *                        TBPTT window of T_w steps runs T_w forwards into slots 0..T_w-1, then the backwards in reverse, handing
 *                        d_mem_in of step t to step t-1 as part of its d_mem_out -- rnn/utils.py:1200-1377's window).
 * csa_phys_train_forward as csa_phys_forward (same arguments), keeping the GRU gates / hidden sequences / head-GEMM output.
 * csa_phys_train_backward given dLoss/d(out_lev) (B,60,5), dLoss/d(out_sfc) (B,8), dLoss/d(mem_out) (B,50,16) and the forward's own
 *                        inputs: grads (nparam floats) += dLoss/dparams, d_mem_in (B,50,16) = dLoss/d(rnn_mem).  Deterministic
 *                        (fixed-order partial sums, no atomics).
 * csa_phys_train_adam_step torch.optim.Adam update of the flat vector (the reference's default optimiser, :678; weight_decay is Adam's L2
 *                        term), then the kernel layouts re-packed (one gather launch).
 * get / set_params copy the flat vector (device pointers).  The inference entry points of the same handle keep the weights given at
 * create: build a new handle from the trained state_dict to serve it. */
int csa_phys_train_enable(csa_phys *h, int nslots);
int csa_phys_train_num_params(csa_phys *h, int *n_tensors, int *n_floats);
int csa_phys_train_param_info(csa_phys *h, int i, const char **name, int *offset, int *rows, int *cols);
int csa_phys_train_get_params(csa_phys *h, float *dst, void *stream);
int csa_phys_train_set_params(csa_phys *h, const float *src, void *stream);
int csa_phys_train_forward(csa_phys *h, int slot, int B, const float *x_main, const float *x_sfc, const float *rnn_mem,
                           const float *x_denorm, int nxd, const float *hx2, float *out_lev, float *out_sfc, float *mem_out, void *stream);
int csa_phys_train_backward(csa_phys *h, int slot, int B, const float *x_main, const float *x_sfc, const float *rnn_mem, const float *x_denorm,
                            int nxd, const float *d_out, const float *d_out_sfc, const float *d_mem_out, float *d_mem_in,
                            float *grads, void *stream);
/* the reference trainer's loss on this model's outputs (huber + energy + water closures, rnn/utils.py:1203-1366, rnn/metrics.py:142-315)
 * with its analytic gradient: arguments as csa_train_loss (pred (Tw*B,60,5) / tgt normalised, yto (Tw*B,60,6) / yto_sfc physical
 * targets, x_raw (Tw*B,60,nxd) = inputs_denorm, x_sfc_n (Tw*B,naux) normalised; scalars: 7 device floats); Tw <= nslots */
int csa_phys_train_loss(csa_phys *h, int B, int Tw, int nxd, float w_energy, float w_water, const float *pred, const float *pred_sfc,
                        const float *tgt, const float *tgt_sfc, const float *yto, const float *yto_sfc, const float *x_raw,
                        const float *x_sfc_n, float *scalars, float *d_pred, float *d_pred_sfc, void *stream);
int csa_phys_train_adam_step(csa_phys *h, const float *grads, float lr, float beta1, float beta2, float eps, float weight_decay,
                             void *stream);

/* ---- stochastic recurrent layers ---------------------------------------------------
 * MyStochasticGRULayer5  rnn/models_torch_kernels.py:834-891 (its GPU path = the repo's inline CUDA, :29-252)
 * MyStochasticLSTMLayer4 rnn/models_torch_kernels.py:1474-1531
 * Weights are HOST pointers in the reference's (in_features, out_features) layout; x is (T,B,nx) sequence-first,
 * eps (T,B,H) is the N(0,1) draw the reference makes at the top of forward, passed explicitly.  max_rows >= T*B. */
typedef struct csa_stoch csa_stoch;
int csa_stoch_gru5_create(int nx, int nh, const float *weight_ih, const float *weight_zh, const float *weight_encoder,
                          const float *bias_ih /* nullable */, const float *bias_zh /* nullable */, int max_rows,
                          csa_stoch **out);
int csa_stoch_lstm4_create(int nx, int nh, const float *weight_encoder, int max_rows, csa_stoch **out);
int csa_stoch_destroy(csa_stoch *h);
int csa_stoch_gru5_forward(csa_stoch *h, int T, int B, const float *x, const float *h0, const float *eps,
                           float *out, void *stream);
int csa_stoch_lstm4_forward(csa_stoch *h, int T, int B, const float *x, const float *h0, const float *c0,
                            const float *eps, float *out, float *hT, float *cT, void *stream);
/* Backward of the two layers.  MyStochasticGRULayer5: the reference's hand-written backward, stochastic_gru_backward_elem_kernel
 * rnn/models_torch_kernels.py:85-126 + the reverse sequence loop :176-232 + FusedCUDAStochasticGRUSequence.backward :826-841;
 * MyStochasticLSTMLayer4: autograd through :1494-1531.  csa_stoch_enable_training allocates the saved-activation buffers (once);
 * *_forward_train is *_forward that also saves them; *_backward consumes them: ONE launch walks the levels backwards with the
 * transposed recurrent weights register-stationary, then one GEMM per weight / input gradient.  d_out (T,B,H) is the gradient of
 * the output sequence (d_hT, d_cT (B,H): of the returned final state, nullable); d_x (T,B,nx), d_h0 / d_c0 (B,H), d_eps (T,B,H,
 * nullable) are written; `grads` is a flat device buffer of csa_stoch_num_params floats, ACCUMULATED into, in the layout
 * GRU5: weight_ih (nx,3H) | weight_zh (H,3H) | weight_encoder (H,2H) [| bias_ih (3H) | bias_zh (3H)], LSTM4: weight_encoder. */
int csa_stoch_enable_training(csa_stoch *h);
long csa_stoch_num_params(const csa_stoch *h);
int csa_stoch_gru5_forward_train(csa_stoch *h, int T, int B, const float *x, const float *h0, const float *eps,
                                 float *out, void *stream);
int csa_stoch_lstm4_forward_train(csa_stoch *h, int T, int B, const float *x, const float *h0, const float *c0,
                                  const float *eps, float *out, float *hT, float *cT, void *stream);
/* The activations *_forward_train saves and *_backward consumes (and overwrites) live in the handle's own buffers unless the caller
 * supplies its own: `acts` = one device buffer of csa_stoch_activation_floats(h, T, B) floats per forward that is still awaiting its
 * backward (the autograd layer of rnn/models_torch_kernels.py:795-841 saves them per call in ctx), NULL = back to the handle's own.
 * Set before *_forward_train and again, with the same buffer, before the matching *_backward. */
long csa_stoch_activation_floats(const csa_stoch *h, int T, int B);
int csa_stoch_set_activations(csa_stoch *h, float *acts, int T, int B);
int csa_stoch_gru5_backward(csa_stoch *h, int T, int B, const float *x, const float *eps, const float *d_out,
                            float *d_x, float *d_h0, float *d_eps, float *grads, void *stream);
int csa_stoch_lstm4_backward(csa_stoch *h, int T, int B, const float *x, const float *eps, const float *d_out,
                             const float *d_hT, const float *d_cT, float *d_x, float *d_h0, float *d_c0, float *d_eps,
                             float *grads, void *stream);

const char *csa_last_error(void);
const char *csa_version(void);

#ifdef __cplusplus
}
#endif
#endif
