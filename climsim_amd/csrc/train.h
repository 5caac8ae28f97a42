// Internal declarations of the training-step kernels (train_rec.hip, train_misc.hip).
#pragma once
#include "common.h"

size_t bwd_rec_packed_floats(int nh);
void bwd_rec_pack_weights(int nh, const float *w_hh, float *packed);
int launch_bwd_rec(int nh, const float *wt_packed, float *GP, const float *Cseq, const float *dH,
                   float *dh0, float *dc0, int B, int L, int rev, hipStream_t s);

size_t bwd_rec_packed_floats_gru(int nh);
void bwd_rec_pack_weights_gru(int nh, const float *w_hh, float *packed);
int launch_bwd_rec_gru(int nh, const float *wt_packed, float *GP, const float *Hseq, const float *dH, float *dh0, int B, int L,
                       int rev, hipStream_t s);

int launch_gemm_tn_partial(const float *A, int lda, const float *Bm, int ldb, float *Cpart, int M, int N1, int N2,
                           int nsplit, hipStream_t s);
#define TN_MAX_SEGS 8
struct TnSegs { const float *A[TN_MAX_SEGS]; const float *B[TN_MAX_SEGS]; int n; };
int launch_gemm_tn_segs(const TnSegs &segs, int lda, int ldb, float *Cpart, int M, int N1, int N2,
                        int nsplit, int conv_L, int conv_cin, hipStream_t s, float *Csum = nullptr);
int launch_gemm_tn_partial_cs(const float *A, int lda, const float *Bm, int ldb, float *Cpart, float *Csum, int M, int N1, int N2,
                              int nsplit, hipStream_t s);
int launch_gemm_tn_conv(const float *A, int lda, const float *Bm, int ldb, float *Cpart, int M, int N1, int N2,
                        int nsplit, int conv_L, int conv_cin, hipStream_t s);
int launch_colsum_partial(const float *A, float *part, int M, int N, int nsplit, hipStream_t s);
int launch_reduce_partials(const float *part, int nsplit, int n, const int *map, const int *map2, float *grad, hipStream_t s);
// several reductions in one launch (train_misc.hip): queue jobs whose targets do not overlap, each with its own partial buffer
#define REDUCE_MAX_JOBS 16
struct ReduceJob { const float *part; const int *map, *map2; int nsplit, n, blk0, small; };
struct ReduceJobs { int n = 0, nblk = 0; ReduceJob j[REDUCE_MAX_JOBS]; };
int reduce_queue_add(ReduceJobs &J, const float *part, int nsplit, int n, const int *map, const int *map2);
int launch_reduce_queue(ReduceJobs &J, float *grad, hipStream_t s);
int reduce_queue_add_2stage(ReduceJobs &J, const float *part, int nsplit, int n, const int *map, const int *map2, float *tmp, int chunks,
                            hipStream_t s);
int launch_reduce_partials_2stage(const float *part, int nsplit, int n, const int *map, const int *map2, float *grad, float *tmp,
                                  int chunks, hipStream_t s);

int head_bwd_partial_floats(const csa_config &c);
int launch_head_bwd(const DevModel &m, int B, const float *d_out, const float *d_out_sfc, const float *d_mem_out,
                    const float *Z, const float *H2, float *dH2, float *part, hipStream_t s);
__host__ __device__ inline int prep_bwd_partial_floats(const csa_config &c)
{
    return c.nh1 * (c.nx + 1) + c.nh1 + 2 * (c.nh1 * c.nx_sfc + c.nh1) + 2 * (c.nh2 * 2 + c.nh2);
}
int launch_prep_bwd(const DevModel &m, int B, const float *dX1, const float *X1, const float *X16, const float *xs_n,
                    const float *hc0, const float *dhc1, const float *dhc2, float *d_mem_in, float *part, hipStream_t s);

int launch_loss(const DevModel &m, const float *hyai, const float *hybi, int B, int Tw, float w_h, float w_w,
                const float *pred, const float *pred_sfc, const float *tgt, const float *tgt_sfc, const float *yto,
                const float *yto_sfc, const float *x_raw, const float *sp, float *samp, float *ecoef, float *scal,
                float *d_pred, float *d_pred_sfc, hipStream_t s);
int launch_adam(float *p, const float *g, float *m1, float *m2, int n, float lr, float b1, float b2, float eps, int step,
                float wd, hipStream_t s);
struct GatherEntry { float *dst; const int *idx; const int *idx2; int n; };
int launch_gather_multi(const GatherEntry *tab_dev, int count, int max_n, const float *src, hipStream_t s);
int launch_gather(float *dst, const float *src, const int *idx, const int *idx2, int n, hipStream_t s);

// prep.hip, training variant: also saves X16 (B,L,nx+1) and the normalised surface inputs (B,nx_sfc)
int launch_prep_train(const DevModel &m, int B, int normalised, const float *x_main, const float *x_sfc,
                      const float *mem_in, float *X1, float *hc0, float *X16, float *xs_n, hipStream_t s);
