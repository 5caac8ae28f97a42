/*
 * climsim_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar fp32 CPU restatement of the reference's per-column emulator path
 * (peterukk/ClimSim rnn/ v4 wrappers).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (climsim_amd/) never links, imports or falls back to it.
 *
 * Parity pinning: tests/test_oracle_golden.py checks this code against golden
 * vectors generated from the reference's own shipped TorchScript artefacts
 * (tests/golden/make_golden.py) -- "parity pinned by reference outputs".
 *
 * Reference call sites restated here (paths relative to /root/reference):
 *   wrapper preprocessing      rnn/save_wrapper_mem.py:411-457, rnn/utils.py:182-217
 *   LayerPressure              rnn/layers.py:117-121
 *   model forward (current)    rnn/models/models.py:432-608
 *   model forward (legacy)     TorchScript code embedded in rnn/v4_rnn*_wrapper*.pt
 *   nn.LSTM / nn.GRU cells     PyTorch ATen semantics (gate order i,f,g,o / r,z,n)
 *   postprocessing / packing   rnn/models/models.py:273-339, rnn/save_wrapper_mem.py:470-497,539
 */
#ifndef CLIMSIM_ORACLE_H
#define CLIMSIM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    /* sizes */
    int nlev;       /* 60 */
    int nx;         /* level inputs, 15 */
    int nx_sfc;     /* 19 */
    int ny;         /* model level outputs (5 for mp_mode 1) */
    int ny_sfc;     /* 8 */
    int nh1, nh2;   /* hidden sizes of rnn1 / rnn2 */
    int nh_mem;     /* 16, or 0 for the stateless model (no mlp_latent) */
    /* behaviour flags */
    int use_lstm;       /* 1 LSTM, 0 GRU */
    int legacy;         /* 1: shipped .pt generation (mem in sequence order, c0=tanh, noise init);
                           0: current RNN_autoreg (mem in level order, c0 linear, mlp_toa init) */
    int output_prune;   /* zero out[lev<12, var>=1]  (models.py:554-559) */
    int mp_mode;        /* 1: diagnose liq/ice from T (wrapper mp_postprocessing); 0: none */
    int snowhice_fix;   /* x_sfc >= 1e10 -> -1 */
    int qinput_prune;   /* x[:,0:15,2] = 0 after normalisation */
    int rh_prune;       /* clamp x[:,:,1] to [0,1.2] */
    int scrub_inf;      /* Inf -> 0 after normalisation (current wrappers); legacy artefacts only scrub NaN */
    int scrub_out_nan;  /* NaN -> 0 on the packed output (save_wrapper_mem.py:539) */
    int q_input_mode;   /* rnn/utils.py:262-272: 0 none, 1 append q from (RH,T,p) as last input, 2 replace RH by q */
    int v5_input;       /* rnn/utils.py:186-198: input 2 = 1-exp(-(qliq+qice) lbd_qn) (pruned BEFORE the transform), input 3 =
                           hardtanh((T-253.16)*0.05, 0, 1) (models.py:260-266); then the usual normalisation */
    /* constants */
    const float *xmean_lev, *xdiv_lev;   /* (nlev,nx) */
    const float *xmean_sca, *xdiv_sca;   /* (nx_sfc) */
    const float *lbd_qc, *lbd_qi;        /* (nlev) */
    const float *lbd_qn;                 /* (nlev), v5_input only */
    const float *yscale_lev;             /* (nlev,ny) */
    const float *yscale_sca;             /* (ny_sfc) */
    const float *hyam, *hybm;            /* (nlev) */
    /* weights, PyTorch layout (out_features, in_features) row-major */
    const float *mlp_initial_w, *mlp_initial_b;       /* (nh1, nx+1) */
    const float *mlp_surface1_w, *mlp_surface1_b;     /* (nh1, nx_sfc) */
    const float *mlp_surface2_w, *mlp_surface2_b;     /* (nh1, nx_sfc)  LSTM only */
    const float *mlp_toa1_w, *mlp_toa1_b;             /* (nh2, 2)  current generation */
    const float *mlp_toa2_w, *mlp_toa2_b;             /* (nh2, 2)  current generation, LSTM */
    const float *rnn1_w_ih, *rnn1_w_hh, *rnn1_b_ih, *rnn1_b_hh;  /* (G*nh1, nh1+nh_mem), (G*nh1, nh1) */
    const float *rnn2_w_ih, *rnn2_w_hh, *rnn2_b_ih, *rnn2_b_hh;  /* (G*nh2, nh1), (G*nh2, nh2) */
    const float *mlp_latent_w, *mlp_latent_b;         /* (nh_mem, nh2)  if nh_mem>0 */
    const float *mlp_output_w, *mlp_output_b;         /* (ny, nh_mem or nh2) */
    const float *mlp_surface_output_w, *mlp_surface_output_b; /* (ny_sfc, nh2) */
} oracle_model;

/* Normalised-space model forward (RNN_autoreg.forward / legacy original_model.forward).
 *   x_main_n (B,nlev,nx), x_sfc_n (B,nx_sfc): already normalised
 *   mem_in: legacy (B,nlev,nh_mem) in sequence order; current (nlev,B,nh_mem) level order; NULL if nh_mem==0
 *   hx2,cx2 (B,nh2): explicit initial state of rnn2 for the legacy generation (NULL otherwise)
 *   out (B,nlev,ny), out_sfc (B,ny_sfc), mem_out same layout as mem_in
 *   rnn1out/rnn2out: optional (B,nlev,nh) level-order debug taps, may be NULL */
int oracle_model_forward(const oracle_model *m, int B,
                         const float *x_main_n, const float *x_sfc_n, const float *mem_in,
                         const float *hx2, const float *cx2,
                         float *out, float *out_sfc, float *mem_out,
                         float *rnn1out, float *rnn2out);

/* Wrapper preprocessing: raw -> normalised (does not touch its inputs). */
int oracle_preprocess(const oracle_model *m, int B, const float *x_main, const float *x_sfc,
                      float *x_main_n, float *x_sfc_n);

/* Full wrapper: raw (B,nlev,nx),(B,nx_sfc)[,(B,nlev,nh_mem)] -> packed yout (B, 368 + nlev*nh_mem)
 * (save_wrapper.py:255-298 / save_wrapper_mem.py:499-545). */
int oracle_wrapper_forward(const oracle_model *m, int B,
                           const float *x_main, const float *x_sfc, const float *mem_in,
                           const float *hx2, const float *cx2, float *yout);

/* Tuple ("ftorch") wrapper of the current generation (rnn/utils.py:260-295):
 * out_lev (B,nlev,6), out_sfc (B,ny_sfc), mem_out (nlev,B,nh_mem). */
int oracle_wrapper_forward_tuple(const oracle_model *m, int B,
                                 const float *x_main, const float *x_sfc, const float *mem_in,
                                 float *out_lev, float *out_sfc, float *mem_out);

#ifdef __cplusplus
}
#endif
#endif
