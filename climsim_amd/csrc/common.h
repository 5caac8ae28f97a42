// Internal definitions shared by the HIP translation units of libclimsim_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/climsim_amd.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Output modes of the head kernel.
enum HeadMode { HEAD_PACKED = 0, HEAD_TUPLE = 1, HEAD_RAW = 2 };

// Device-side view of one model (all pointers are device pointers).
struct DevModel {
    csa_config cfg;
    // level-major memory tensors (current generation, (L, B, nh_mem)) of a sub-batch: row (l, b) of this call sits at
    // (l * mem_B + mem_off + b); mem_B == 0 means the call's own B and no offset (api.hip::run_forward_halves)
    int mem_B, mem_off;
    // constants
    const float *xmean_lev, *xdiv_lev, *xmean_sca, *xdiv_sca, *lbd_qc, *lbd_qi;
    const float *lbd_qn;                // v5_input only
    const float *yscale_lev, *yscale_sca, *hyam, *hybm;
    // small MLPs, transposed to (in, out) so that thread j reads column j coalesced
    const float *init_wt, *init_b;      // (nx+1, nh1)
    const float *s1_wt, *s1_b;          // (nx_sfc, nh1)
    const float *s2_wt, *s2_b;          // (nx_sfc, nh1)
    const float *toa1_wt, *toa1_b;      // (2, nh2)
    const float *toa2_wt, *toa2_b;      // (2, nh2)
    // input projections: rows permuted to unit-major gate order n' = u*G + g
    const float *wih1, *bias1;          // (G*nh1, nh1+nh_mem), (G*nh1)   bias = b_ih + b_hh (GRU: b_hn kept apart)
    const float *wih2, *bias2;          // (G*nh2, nh1)
    const float *bhn1, *bhn2;           // GRU only: b_hn (nh)
    // recurrent weights packed for the register-stationary kernel (see rec.hip)
    const float *whh1p, *whh2p;
    const float *whh1q, *whh2q;         // the same matrices packed for the one-column kernel (lstm_rec1_kernel), LSTM only
    const float *whh1g, *whh2g;         // GRU, nh <= 128: packed for the second-generation two-column kernel (gru_rec2_kernel)
    const float *whh1m, *whh2m;         // LSTM, nh <= 128: k-major packing of the matrix-pipe four-column kernel (lstm_rec4m_kernel)
    // heads
    const float *lat_wt, *lat_b;        // (nh2, nh_mem)
    const float *out_w, *out_b;         // (ny, nh_mem or nh2) row-major
    const float *out_wt;                // (nh2, ny) transposed copy, stateless model only
    const float *sfo_w, *sfo_b;         // (ny_sfc, nh2) row-major
};

#define CSA_HIP_CHECK(expr)                                                        \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            csa_set_error(#expr, e_);                                              \
            return CSA_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

// raise a kernel's dynamic-LDS limit once per device (kernels that keep part of their weights in LDS, nh = 144)
#define CSA_SET_DYN_LDS_ONCE(kern, bytes)                                                                          \
    do {                                                                                                           \
        static unsigned long long done_ = 0;                                                                       \
        int dev_ = 0;                                                                                              \
        (void)hipGetDevice(&dev_);                                                                                 \
        if (!((done_ >> (dev_ & 63)) & 1ull)) {                                                                    \
            CSA_HIP_CHECK(hipFuncSetAttribute((const void *)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            done_ |= 1ull << (dev_ & 63);                                                                          \
        }                                                                                                          \
    } while (0)

void csa_set_error(const char *what, hipError_t e);
void csa_set_error_msg(const char *msg);

// ---- kernel launchers (each returns a csa_status) ---------------------------------------
// prep.hip: normalise + pressure + mlp_initial + memory concat -> X1 (L,B,nh1+nh_mem) sequence
// order; surface / TOA MLPs -> h0c0 (4,B,nh): [h0_rnn1, c0_rnn1, h0_rnn2, c0_rnn2].
int launch_prep(const DevModel &m, int B, int normalised, const float *x_main, const float *x_sfc,
                const float *mem_in, const float *hx2, const float *cx2,
                float *X1, float *hc0, hipStream_t s);

// gemm.hip: C(M,N) = A(M,K) * W(N,K)^T + bias(N), fp32 MFMA.
int launch_proj_gemm(const float *A, const float *W, const float *bias, float *C,
                     int M, int N, int K, hipStream_t s, int class_rows = 0);

// same GEMM with a fused activation epilogue (MLP baseline): act 0 none, 1 LeakyReLU(alpha), 2 split linear|ReLU head
int launch_gemm_act(const float *A, const float *W, const float *bias, float *C, int M, int N, int K,
                    int act, float alpha, int n_lin, hipStream_t s);

// optional epilogue extras of the general GEMM (all null / zero = none); see gemm.hip
struct GemmEpi {
    const float *addsrc = nullptr;          // v += addsrc[row*ldc + col]
    const unsigned char *mask = nullptr;    // v *= mask[row*ldmask + col] ? mscale : 0   (col < ldmask)
    int ldmask = 0;
    float mscale = 1.0f;
    const float *gate = nullptr;            // v *= gate[row*ldc + col] > 0 ? gscale : gneg
    float gscale = 1.0f, gneg = 0.0f;       // (ReLU + dropout backward: gneg 0; LeakyReLU(alpha) backward: gscale 1, gneg alpha)
};
int launch_gemm_epi(const float *A, const float *W, const float *bias, float *C, int M, int N, int K, int act, float alpha,
                    int n_lin, int lda, int ldc, int conv_L, int conv_cin, const GemmEpi &epi, hipStream_t s);

// small-M variant: 32x32 tile per workgroup, K split over its four waves (gemm.hip)
int launch_gemm_small(const float *A, const float *W, const float *bias, float *C, int M, int N, int K, int act, float alpha,
                      int n_lin, hipStream_t s);

// general form: leading dimensions, implicit-GEMM conv1d(k=3,'same') over (column, level) rows, C += mode, ELU
int launch_gemm_ex(const float *A, const float *W, const float *bias, float *C, int M, int N, int K, int act, float alpha,
                   int n_lin, int lda, int ldc, int conv_L, int conv_cin, int accumulate, hipStream_t s);

// rec.hip: level-recurrent LSTM/GRU over L steps; P (L,B,G*nh) pre-activations in sequence
// order, Hout (L,B,nh) written at level index (reverse ? L-1-t : t).
int launch_rec(int use_lstm, int nh, const float *whh_packed, const float *bhn, const float *P,
               const float *h0, const float *c0, float *Hout, int B, int L, int reverse_out,
               hipStream_t s);
// training forward: as launch_rec (LSTM) + saves gates in place over P, c_t to Cseq and h_t to Hseq
int launch_rec1_gru(int nh, const float *whh_packed, const float *bhn, const float *P, const float *h0, float *Hout, int B, int L,
                    int reverse_out, hipStream_t s);
// one-column-per-workgroup LSTM kernel (small batches) and its weight packing
void rec1_pack_weights(int nh, const float *w_hh, float *packed);
// second-generation two-column GRU kernel (inference, nh in {64, 96, 128}) and its weight packing
void gru2_pack_weights(int nh, const float *w_hh, float *packed);
int launch_rec2_gru(int nh, const float *whh_g2, const float *bhn, const float *P, const float *h0, float *Hout, int B, int L,
                    int reverse_out, hipStream_t s);
int launch_rec1(int nh, const float *whh_packed1, const float *P, const float *h0, const float *c0, float *Hout, int B, int L,
                int reverse_out, hipStream_t s);
int launch_rec_train(int nh, const float *whh_packed, float *P, const float *h0, const float *c0, float *Hout,
                     int B, int L, int reverse_out, float *Hseq, float *Cseq, hipStream_t s);
int launch_rec_train_gru(int nh, const float *whh_packed, const float *bhn, float *P, const float *h0, float *Hout, int B, int L,
                         int reverse_out, float *Hseq, hipStream_t s);
// matrix-pipe four-column LSTM kernel (large batches): packing, selection (env CSA_REC4_KERNEL / CSA_REC4_MIN_BATCH), launch
void rec4m_pack_weights(int nh, const float *w_hh, float *packed);
int launch_rec4m_train(int nh, const float *whh_m, float *P, const float *h0, const float *c0, float *Hout, int B, int L,
                       int reverse_out, float *Hseq, float *Cseq, hipStream_t s);
void gru4m_pack_weights(int nh, const float *w_hh, float *packed);
int launch_rec4m_gru(int nh, const float *whh_m, const float *bhn, const float *P, const float *h0, float *Hout, int B, int L,
                     int reverse_out, hipStream_t s);
bool gru4m_selected(int nh, int B);
int launch_rec4m_train_gru(int nh, const float *whh_m, const float *bhn, float *P, const float *h0, float *Hout, int B, int L,
                           int reverse_out, float *Hseq, hipStream_t s);
bool rec4m_selected(int use_lstm, int nh, int B);
int launch_rec4m(int nh, const float *whh_m, const float *P, const float *h0, const float *c0, float *Hout, int B, int L,
                 int reverse_out, hipStream_t s);
size_t rec_packed_floats(int use_lstm, int nh);
// host-side packer: W_hh (G*nh, nh) PyTorch layout -> register-stationary layout
void rec_pack_weights(int use_lstm, int nh, const float *w_hh, float *packed);

// head.hip: mlp_latent / mlp_output / surface head / de-normalisation / microphysics / packing
int launch_head(const DevModel &m, int B, int mode, const float *H2, const float *x_main_raw, const float *x_sfc_raw,
                float *y0, float *y1, float *y2, hipStream_t s);

// head.hip: RNN_autoreg.postprocessing(out, out_sfc, x_denorm) on its own (models.py:273-339)
int launch_postprocess(const DevModel &m, int B, const float *out, const float *out_sfc, const float *x_denorm, int nxd,
                       float *out6, float *out_sfc_d, hipStream_t s);
int launch_to_level_major(int B, int L, int n, const float *src, float *dst, hipStream_t s);
int launch_pack_ar(int B, int L, int nys, int nm, int nh, const float *out6, const float *out_sfc, const float *mem,
                   const float *eps, float *yout, hipStream_t s);

// rows of `rowf` floats (a multiple of 4, 16-byte aligned), `stride` floats apart in global memory -> consecutive rows of LDS
// (row pitch `pitch` floats).  All J float4 loads of a thread are issued before the first LDS store: a rolled loop of
// load -> store pairs pays one memory round trip per trip (this was 20-30 us of head_bwd / prep_bwd, three launches each per step).
template <int J, int T>
__device__ __forceinline__ void rows_to_lds(float *dst, int pitch, const float *__restrict__ src, size_t stride, int nrow, int rowf, int tid)
{
    const int q = rowf >> 2, total = nrow * q;
    for (int base = 0; base < total; base += J * T) {
        f32x4 v[J];
        int r[J], c[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int idx = min(base + tid + T * j, total - 1);
            r[j] = idx / q; c[j] = (idx - r[j] * q) * 4;
            v[j] = *(const f32x4 *)(src + (size_t)r[j] * stride + c[j]);
        }
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (base + tid + T * j < total) *(f32x4 *)(dst + r[j] * pitch + c[j]) = v[j];
    }
}

