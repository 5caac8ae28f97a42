// phys.h -- shared state of the physRNN "Hidden" path (phys.hip: core + decoder, phys_rad.hip: radiation scheme)
#pragma once
#include "common.h"
#include "pack.h"
#include <vector>

#define PH_L 60
#define PH_NHEAD 11
#define PH_NG 16            // g-points of the radiation scheme (ng_lw = ng_sw = 16 in every shipped artefact)
#define PH_NRETAB 138

struct PhysDev {
    int nx;                 // columns of x_main
    int nfeat;              // leading columns of x_main that feed mlp_initial (the layer-pressure feature follows them)
    int naux;               // columns of x_sfc
    int nx_sfc;             // inputs of mlp_surface1: aux columns [0, sfc_cut) and [sfc_cut + sfc_skip, naux)
    int sfc_cut, sfc_skip;
    int nh, ilev, nm0, Lc;
    int ltop, Lr;           // first level the GRUs see and their sequence length (0 / 60, or ilev / 50)
    int ncol, hdw;          // mp_ncol; width of the head GEMM
    int rad;                // 1: physical radiation scheme (no mlp_output_rad / mlp_surface_output_rad heads)
    int liq_off;            // column of the mlp_liq_frac_crm head in the head GEMM, or -1: liquid fraction from temperature
    int physrad;            // the physRNN_physRad-* graphs: clear-sky region 0 and no sub-grid temperature in the decoder, latent heating
                            // from area-summed rates, vapour mixing ratio q / (1 - q), rnn_mem level-major (50, B, 16) in and out
    const float *hyam, *hybm, *hyai, *hybi, *yscale_lev, *yscale_sca;
    float xdiv_sca0, xmean_sca0;
    const float *init_wt, *init_b, *s1_wt, *s1_b;   // (nfeat+1, nh), (nx_sfc, nh) transposed
    const float *out_w, *out_b;                     // mlp_output (5, nm0)
    const float *sfo_w, *sfo_b;                     // mlp_surface_output_rad (6, nh)            (rad == 0)
    const float *rel_w, *rel_b;                     // mlp_precip_release (1, nh)
    // radiation scheme (rad == 1)
    const float *xmean_sca, *xdiv_sca, *lbd_qn, *g_xmin, *g_range, *g_ymean, *g_ystd, *ys_rad, *toa_spec, *retab;
    const float *cld_w, *cld_b;   // cloud_optics_lw (16, 19), (16): learned cloud LW optical depth per unit path (num88955), or null
    int lw_dn;              // 1: the LW downward sweep gets its own source (later exports); 0: the upward one, as first serialised
    // physics_rad_e3sm generation (num94634): SW optical properties from two gas-optics MLPs + Slingo / Ebert-Curry cloud optics
    const float *swg;       // the packed block of include/climsim_amd.h (CSA_PHYS_SW_GAS), or null: SW head MLP
    const float *cld_sw_w, *cld_sw_b;   // learned SW cloud optics (48, 19), (48): cloud_optics_sw2 o cloud_optics_sw composed (num88741), or null
    // the "nx21" generation of the frozen `*_wrapped` exports (csa_phys_wrapped_create):
    int memlm;              // rnn_mem level-major (50, B, 16) in and out (physrad graphs and nx21)
    int mem_B, mem_off;     // level-major caller tensors (rnn_mem, mem_out, mask_u) of a column SUB-RANGE of the call: row (l, b) sits at
                            // l * mem_B + mem_off + b; mem_B == 0: the call's own B, no offset (csa_phys_wrapped_forward's column halves)
    int nx21;               // decoder: eddy heat flux zero at the surface, per-region liquid fraction shared with the cloud optics; radiation:
                            // vapour mixing ratio q / (1 - q), 7-32-32-16 SW gas optics with the humidity coin (swg = the SWX_* block), ice SW
                            // optics on the LIQUID radius (as serialised), no relu on the SW fluxes, NET shortwave as first surface output
    int rad_qv_upd;         // 1: radiation reads the updated grid-mean q_v, 0: the un-updated one
    int n_ir, n_mix;        // g-points [0, n_ir) near-infrared, [n_ir, n_mix) mixed, the rest visible (surface albedo, SOLL / SOLS split)
    float mix_near, mix_vis;   // weights of the mixed g-points (learned in most variants; 0.5 / 0.5 otherwise and in the older graphs)
    const float *cldtab;    // (12, 16): Slingo A..F then Ebert-Curry a..f per g-point; cld_band: (12, 4) per BAND, then the (4, 16) band -> g-point matrix
    int cld_band;           // 1: later exports (num32701, num87824): k, k ssa, k ssa g of the four bands times a learned band -> g-point matrix
    int ice_re;             // 1: the ice SW optics read the ICE effective radius (later exports); 0: the liquid one clamped to 13..130, as first serialised
    int gridT, clear0;      // decoder: no sub-grid temperature (grid temperature in the eddy flux and the liquid ramp, latent heating from the
                            // area-summed rates) / region 0 holds no condensate.  Both = physrad in the unfrozen graphs; the exports num45826 /
                            // num74834 have the first without the second
    int dec21;              // nx21-style decoder (heat flux at layer tops, zero at the surface; per-region liquid fraction): nx21 && !gridT
    int cld_qn_old;         // 1: the radiation scheme's cloud water paths take the sub-grid cloud water BEFORE the step (num45826 / num74834)
    int sw_e3sm;            // nx21 wrapper + solver around the unfrozen physics_rad_e3sm SW gas optics (112 k-points, mean of the two humidity
                            // variants, reductions): swg = the SWG_* block, cldtab = the cloud table / band matrix
    int rad_T_old;          // nx21: 1 = the radiation scheme reads the temperature BEFORE the step (num36398)
    int rad_qn_upd;         // SW head MLP of the earlier nx21 exports: 1 = it sees the UPDATED grid-mean cloud water
    int sfc_sw_down;        // nx21: 1 = the first surface output is the DOWNWARD shortwave (num82174), 0 = the net one
    int cld_liq_upd;        // nx21: 1 = cloud LW optics take the liquid fraction of the UPDATED sub-grid temperature (num82174)
    int sw_ngk;             // > 0: the SW gas models give sw_ngk k-points, reduced to the 16 g-points by Linear + softplus * 0.01 BEHIND the humidity coin
};

// layout of the nx21 SW gas-optics block (floats): input range (same offsets as SWG_XMIN / SWG_XDIV), then per model (absorption,
// Rayleigh) W1 (32, 8: 7 inputs, zero-padded), b1, W2 (32, 32), b2, W3 (16, 32: ng rows, zero-padded), b3
#define SWX_MODEL0 16
#define SWX_W1 0
#define SWX_B1 (SWX_W1 + 32 * 8)
#define SWX_W2 (SWX_B1 + 32)
#define SWX_B2 (SWX_W2 + 32 * 32)
#define SWX_W3 (SWX_B2 + 32)
#define SWX_B3 (SWX_W3 + 16 * 32)
#define SWX_MODEL_FLOATS (SWX_B3 + 16)
#define SWX_RED (SWX_MODEL0 + 2 * SWX_MODEL_FLOATS)     // k-point reductions (num11916, num87824): W (16, 16: 16 g rows, k zero-padded), b (16), absorption then Rayleigh
#define SWX_RED_FLOATS (16 * 16 + 16)
#define SWX_FLOATS (SWX_RED + 2 * SWX_RED_FLOATS)

// layout of the CSA_PHYS_SW_GAS block (floats): input range, two gas-optics models, the two 112 -> 16 reductions, cloud-optics
// coefficients per g-point.  The 112-wide axis is zero-padded to 128 (four 32-column MFMA tiles; a padded k-point has
// ystd = ymean = 0, i.e. optical depth 0, and zero reduction weights).
#define SWG_NK 112
#define SWG_NKP 128
#define SWG_LD1 12                             // row strides (floats) of the weight matrices: K + 4, so that the 32 lanes reading
#define SWG_LDK 36                             // weight rows n = 0..31 as float4 (MFMA B operand) spread over all LDS banks
#define SWG_LDR 132
#define SWG_XMIN 0
#define SWG_XDIV 8
#define SWG_MODEL0 16
#define SWG_W1 0                               // (32, 8 of 12): 7 inputs, zero-padded
#define SWG_B1 (SWG_W1 + 32 * SWG_LD1)
#define SWG_W2 (SWG_B1 + 32)                   // (32, 32 of 36)
#define SWG_B2 (SWG_W2 + 32 * SWG_LDK)
#define SWG_W3 (SWG_B2 + 32)                   // (128, 32 of 36)
#define SWG_B3 (SWG_W3 + SWG_NKP * SWG_LDK)
#define SWG_YSTD (SWG_B3 + SWG_NKP)
#define SWG_YMEAN (SWG_YSTD + SWG_NKP)
#define SWG_MODEL_FLOATS (SWG_YMEAN + SWG_NKP)
#define SWG_RED (SWG_MODEL0 + 2 * SWG_MODEL_FLOATS)   // reduce1 (16, 128 of 132), bias (16), reduce2 likewise
#define SWG_RED_FLOATS (16 * SWG_LDR + 16)
#define SWG_CLD (SWG_RED + 2 * SWG_RED_FLOATS)        // (12, 16): Slingo A..F then Ebert-Curry a..f, per g-point
#define SWG_FLOATS (SWG_CLD + 12 * 16)

struct csa_phys {
    PhysDev d;
    int max_batch;
    float *wih1, *bias1, *bhn1, *whh1p, *whh1g, *wih2, *bias2, *bhn2, *whh2p, *whh2g, *whead, *bhead;
    float *whh1m = nullptr, *whh2m = nullptr;     // gru_rec4m_kernel layout
    float *X1, *P, *H1, *H2, *hx, *HD;
    // radiation scheme: MLP weights (row-major (out, in), K padded to a multiple of 4) and per-call work arrays
    float *g_w1, *g_b1, *g_w2, *g_b2, *g_w3, *g_b3, *r1_w, *r1_b, *r2_w, *r2_b, *s1_w, *s1_b, *s2_w, *s2_b;
    float *XG, *XR, *RS, *CL, *TP, *S2;
    float *CS = nullptr;    // SW_GAS: cloud SW extinction, scattering, asymmetry per (CRM level, column, g-point): (Lc * B, 48)
    // add_stochastic_layer graphs: rnn3 (MyStochasticGRULayer5 over rnn2's output), its output and the perturbed sequence
    struct csa_stoch *rnn3 = nullptr;
    float *H3 = nullptr, *H2p = nullptr;
    int rnn3_last_mul = 0;  // the release / surface heads read rnn2's last state TIMES the third RNN's (num36398) instead of the latter alone
    // frozen `*_wrapped` exports: the wrapper's constants (rnn/utils.py:182-217) and the buffers between its two passes and the model
    int ng = PH_NG;         // g-points of the export (= nreg: 12 / 14 / 16; the kernels run 16 with zero-weight padding)
    float *wr_xmean = nullptr, *wr_xdiv = nullptr, *wr_lqc = nullptr, *wr_lqi = nullptr;
    float *XM = nullptr, *XS = nullptr, *XD = nullptr, *O5 = nullptr, *OS = nullptr;
    // training (phys_train.hip; non-radiative graph): the trainable tensors as given at create, and the state csa_phys_train_enable builds
    // column halves of csa_phys_wrapped_forward (from CSA_PHYS_HALVES_MIN columns): second half on a side stream, one fork + one join
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<float> host_params;
    struct PhysTrain *tr = nullptr;
    std::vector<void *> owned;
};

#define PH_XG_K 24          // 18 gas-optics inputs, zero-padded to a multiple of 8 (two k-quads per MFMA group)
#define PH_XR_K 24

// head-GEMM column order
enum { H_QV = 0, H_QN, H_T, H_AREA, H_FLUX, H_EDDY, H_QICE, H_SED, H_EVAP, H_COND, H_AA };

#ifdef __HIPCC__
__device__ __forceinline__ float ph_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }   // torch.softplus(beta 1, threshold 20)
template <int N> __device__ __forceinline__ float ph_sum(float v)
{
#pragma unroll
    for (int o = N / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int N> __device__ __forceinline__ float ph_max(float v)
{
#pragma unroll
    for (int o = N / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// row of (column b, CRM level l) in rnn_mem / mem_out: (B, 50, 16), or level-major (50, B, 16) for the physRad graphs
__device__ __forceinline__ size_t ph_mem_row(const PhysDev &d, int B, int b, int l)
{
    return d.memlm ? (size_t)l * (d.mem_B ? d.mem_B : B) + d.mem_off + b : (size_t)b * d.Lc + l;
}

#endif

// phys.hip: the two kernels the training forward shares with inference
int launch_phys_prep(const PhysDev &d, int B, const float *x_main, const float *x_sfc, const float *mem, float *X1, float *hx, hipStream_t s);
int launch_phys_decode_hidden(const PhysDev &d, int B, const float *HD, const float *Hlast, const float *x_sfc, const float *mem,
                              const float *x_denorm, int nxd, float *out_lev, float *out_sfc, float *mem_out, hipStream_t s);
// phys_train.hip
void phys_train_free(struct PhysTrain *t);

// phys_rad.hip
int launch_phys_radiation(csa_phys *h, int B, const float *x_sfc, float *out_lev, float *out_sfc, hipStream_t s, const float *mask_u = nullptr);
