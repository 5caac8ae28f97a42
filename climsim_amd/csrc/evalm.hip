// evalm.hip -- the evaluation scores of the reference's data_utils, computed where the
// predictions already are:
//   climsim_utils/data_utils.py:1843-1857 calc_MAE   mean_t |p - y|
//   climsim_utils/data_utils.py:1859-1874 calc_RMSE  sqrt(mean_t (p - y)^2)
//   climsim_utils/data_utils.py:1876-1892 calc_R2    1 - sum_t (p - y)^2 / sum_t (y - mean_t y)^2
//   climsim_utils/data_utils.py:1894-1908 calc_bias  mean_t p - mean_t y
//   climsim_utils/data_utils.py:1910-1935 calc_CRPS  mean_{t,s} |x_s - y| - mean_t sum_k (x_(k+1) - x_(k)) k (S-k) / (S (S-1))
// Inputs are (time, grid, level) row-major (scalars: level = 1): the reductions run over TIME for each (grid, level)
// cell, optionally followed by the reference's avg_grid mean over grid.  All three kernels are HBM-bound single passes:
//   eval_partial_kernel  one thread per cell c = g*L + l (consecutive lanes -> consecutive addresses), the time axis
//                        split over blockIdx.y; five float64 sums per (split, cell).  The target's variance uses the
//                        shifted form sum (y - y0)^2 - (sum (y - y0))^2 / T with y0 = y[t = 0, c]: one pass instead of the
//                        reference's two, without the cancellation of the unshifted formula.
//   eval_final_kernel    fixed-order sum over the splits -> the four scores per cell; avg_grid: one more fixed-order
//                        mean over grid per level.
//   eval_crps_kernel     one thread per (t, cell): its S members staged in LDS ([member][thread], conflict-free), the
//                        sorted-difference sum of the reference evaluated as sum_{i<j} |x_i - x_j| (identical value, no sort).
#include "common.h"

#define EV_SUMS 5
#define EV_LD 129

__global__ __launch_bounds__(256) void eval_partial_kernel(const float *__restrict__ pred, const float *__restrict__ targ,
                                                           double *__restrict__ part, int T, int C, int tsplit)
{
    const int c = blockIdx.x * 256 + threadIdx.x, sp = blockIdx.y;
    if (c >= C) return;
    const int t0 = (int)((long)T * sp / tsplit), t1 = (int)((long)T * (sp + 1) / tsplit);
    const float y0 = targ[c];
    double sa = 0.0, sq = 0.0, spd = 0.0, sy = 0.0, syy = 0.0;
    auto acc = [&](float p, float y) {
        const float d = p - y;                       // float32 difference, as numpy takes it on float32 arrays
        sa += fabsf(d);
        sq += (double)d * d;
        spd += p;
        const double ys = (double)y - (double)y0;
        sy += ys;
        syy += ys * ys;
    };
    int t = t0;
    for (; t + 4 <= t1; t += 4) {                    // eight independent loads in flight per lane
        float p[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { p[u] = pred[(size_t)(t + u) * C + c]; y[u] = targ[(size_t)(t + u) * C + c]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc(p[u], y[u]);
    }
    for (; t < t1; ++t) acc(pred[(size_t)t * C + c], targ[(size_t)t * C + c]);
    double *o = part + ((size_t)sp * C + c) * EV_SUMS;
    o[0] = sa; o[1] = sq; o[2] = spd; o[3] = sy; o[4] = syy;
}

__global__ __launch_bounds__(256) void eval_final_kernel(const double *__restrict__ part, const float *__restrict__ targ,
                                                         float *__restrict__ out, int T, int C, int tsplit)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s[EV_SUMS] = {0, 0, 0, 0, 0};
#pragma unroll 4
    for (int sp = 0; sp < tsplit; ++sp)
#pragma unroll
        for (int k = 0; k < EV_SUMS; ++k) s[k] += part[((size_t)sp * C + c) * EV_SUMS + k];
    const double n = (double)T, y0 = (double)targ[c];
    const double tss = s[4] - s[3] * s[3] / n;
    out[c] = (float)(s[0] / n);                                 // MAE
    out[(size_t)C + c] = (float)sqrt(s[1] / n);                 // RMSE
    out[2 * (size_t)C + c] = (float)(1.0 - s[1] / tss);         // R2 (tss = 0 -> -inf / nan exactly as numpy's division)
    out[3 * (size_t)C + c] = (float)(s[2] / n - (s[3] / n + y0));   // bias
}

// mean over grid of nrow rows of (G, L) -> (nrow, L); one workgroup per (row, level): lanes stride over grid, float64
// partials, tree reduction in a fixed order (a single thread walking the 384 columns costs 384 dependent L2 latencies = 90 us)
__global__ __launch_bounds__(256) void eval_gridmean_kernel(const float *__restrict__ in, float *__restrict__ out, int nrow, int G, int L)
{
    __shared__ double red[256];
    const int r = blockIdx.x / L, l = blockIdx.x - r * L, tid = threadIdx.x;
    double a = 0.0;
    for (int g = tid; g < G; g += 256) a += in[((size_t)r * G + g) * L + l];
    red[tid] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) out[blockIdx.x] = (float)(red[0] / (double)G);
}

__global__ __launch_bounds__(128) void eval_crps_kernel(const float *__restrict__ sp, const float *__restrict__ targ,
                                                        float *__restrict__ part, long N, int S)
{
    extern __shared__ float xs[];                    // [S][EV_LD]: the odd stride keeps the transposing stores conflict-free too
    const int tid = threadIdx.x;
    const long n = (long)blockIdx.x * 128 + tid;
    // the block's 128*S values are contiguous: read them coalesced, scatter to [member][thread]
    const long base = (long)blockIdx.x * 128 * S, lim = N * S;
    for (int i = tid; i < 128 * S; i += 128) {
        const long e = base + i;
        const int th = i / S, m = i - th * S;
        xs[m * EV_LD + th] = e < lim ? sp[e] : 0.0f;
    }
    __syncthreads();
    if (n >= N) return;
    const float y = targ[n];
    float mae = 0.0f, spread = 0.0f;
    int i = 0;
    for (; i + 4 <= S; i += 4) {                     // four members held in registers per sweep: a quarter of the LDS reads
        const float x0 = xs[i * EV_LD + tid], x1 = xs[(i + 1) * EV_LD + tid], x2 = xs[(i + 2) * EV_LD + tid], x3 = xs[(i + 3) * EV_LD + tid];
        mae += (fabsf(x0 - y) + fabsf(x1 - y)) + (fabsf(x2 - y) + fabsf(x3 - y));
        float a = ((fabsf(x0 - x1) + fabsf(x0 - x2)) + (fabsf(x0 - x3) + fabsf(x1 - x2))) + (fabsf(x1 - x3) + fabsf(x2 - x3));
        for (int j = i + 4; j < S; ++j) {
            const float xj = xs[j * EV_LD + tid];
            a += (fabsf(x0 - xj) + fabsf(x1 - xj)) + (fabsf(x2 - xj) + fabsf(x3 - xj));
        }
        spread += a;
    }
    for (; i < S; ++i) {
        const float xi = xs[i * EV_LD + tid];
        mae += fabsf(xi - y);
        float a = 0.0f;
        for (int j = i + 1; j < S; ++j) a += fabsf(xi - xs[j * EV_LD + tid]);
        spread += a;
    }
    part[2 * n] = mae;
    part[2 * n + 1] = spread;
}

__global__ __launch_bounds__(256) void eval_crps_final_kernel(const float *__restrict__ part, float *__restrict__ out, int T, int C, int S)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double m = 0.0, s = 0.0;
#pragma unroll 8
    for (int t = 0; t < T; ++t) { m += part[2 * ((size_t)t * C + c)]; s += part[2 * ((size_t)t * C + c) + 1]; }
    const double mae = m / ((double)T * S), spread = s / (double)T;
    out[c] = (float)(mae - (S > 1 ? spread / ((double)S * (S - 1)) : 0.0));
}

static int ev_tsplit(int T, int C)
{
    // enough (split, cell) threads to cover the part a few times over, at least 8 time steps per split
    int want = (256 * 256 * 16 + C - 1) / C;
    if (want > T / 8) want = T / 8;
    return want < 1 ? 1 : (want > 64 ? 64 : want);
}

extern "C" long csa_eval_scratch_bytes(int T, int G, int L, int S)
{
    if (T <= 0 || G <= 0 || L <= 0 || S < 0) return -1;
    const long C = (long)G * L;
    const long a = (long)ev_tsplit(T, (int)C) * C * EV_SUMS * (long)sizeof(double) + 4 * C * (long)sizeof(float);
    const long b = S > 0 ? 2 * (long)T * C * (long)sizeof(float) + C * (long)sizeof(float) : 0;
    return a > b ? a : b;
}

// pred, target (T, G, L) device; out: (4, G, L) or with avg_grid (4, L): rows MAE, RMSE, R2, bias
extern "C" int csa_eval_metrics(int T, int G, int L, const float *pred, const float *target, int avg_grid, void *scratch,
                                float *out, void *stream)
{
    if (T <= 0 || G <= 0 || L <= 0 || !pred || !target || !scratch || !out) { csa_set_error_msg("csa_eval_metrics: bad argument"); return CSA_ERR_ARG; }
    if ((long)G * L > 0x7fffffffL / 8) { csa_set_error_msg("csa_eval_metrics: grid x level too large"); return CSA_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    const int C = G * L, ts = ev_tsplit(T, C), gx = (C + 255) / 256;
    double *part = (double *)scratch;
    float *cell = avg_grid ? (float *)(part + (size_t)ts * C * EV_SUMS) : out;
    hipLaunchKernelGGL(eval_partial_kernel, dim3(gx, ts), dim3(256), 0, s, pred, target, part, T, C, ts);
    hipLaunchKernelGGL(eval_final_kernel, dim3(gx), dim3(256), 0, s, part, target, cell, T, C, ts);
    if (avg_grid) hipLaunchKernelGGL(eval_gridmean_kernel, dim3(4 * L), dim3(256), 0, s, cell, out, 4, G, L);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// samplepreds (T, G, L, S), target (T, G, L) device; out (G, L) or with avg_grid (L)
extern "C" int csa_eval_crps(int T, int G, int L, int S, const float *samplepreds, const float *target, int avg_grid,
                             void *scratch, float *out, void *stream)
{
    if (T <= 0 || G <= 0 || L <= 0 || S <= 0 || !samplepreds || !target || !scratch || !out) { csa_set_error_msg("csa_eval_crps: bad argument"); return CSA_ERR_ARG; }
    if (S > 256) { csa_set_error_msg("csa_eval_crps: at most 256 ensemble members"); return CSA_ERR_UNSUPPORTED; }
    if ((long)G * L > 0x7fffffffL / 8) { csa_set_error_msg("csa_eval_crps: grid x level too large"); return CSA_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    const int C = G * L;
    const long N = (long)T * C;
    float *part = (float *)scratch, *cell = avg_grid ? part + 2 * N : out;
    const size_t lds = (size_t)S * EV_LD * sizeof(float);
    CSA_HIP_CHECK(hipFuncSetAttribute((const void *)eval_crps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * EV_LD * (int)sizeof(float)));
    hipLaunchKernelGGL(eval_crps_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), lds, s, samplepreds, target, part, N, S);
    hipLaunchKernelGGL(eval_crps_final_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, cell, T, C, S);
    if (avg_grid) hipLaunchKernelGGL(eval_gridmean_kernel, dim3(L), dim3(256), 0, s, cell, out, 1, G, L);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
