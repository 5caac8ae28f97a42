// crps.hip -- the ensemble score of the reference's probabilistic training, value and gradient:
//   rnn/metrics.py:535-626  CRPS(y, y_sfc, y_pred, y_sfc_pred, timesteps, beta, alpha)
// an energy-score form: with z = [level outputs | surface outputs] (D values per sample),
//   skill  = mean over samples and members of ||z_true - z_e||_2 / sqrt(D)
//   spread = (1 - eps) * sum_{e,e'} mean over samples of ||z_e - z_e'||_2 / (E (E-1)) / sqrt(D),   eps = (1 - alpha) / E
//   CRPS   = 2 beta skill - spread
// Predictions are ordered (time, member, column) as the ensemble forward produces them (rnn/utils.py:1065-1075).
// One workgroup per (time, column) sample: truth and all member vectors staged in LDS, E + E(E-1)/2 distances by
// block reductions, per-sample sums to a buffer; a second single-workgroup pass adds them in a fixed order.
#include "common.h"

__device__ __forceinline__ float crps_block_sum(float v, float *red)
{
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const float r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void crps_sample_kernel(
    const float *__restrict__ y, const float *__restrict__ ys, const float *__restrict__ yp, const float *__restrict__ yps,
    float *__restrict__ part, int B, int E, int D1, int D2)
{
    extern __shared__ float sm[];
    const int D = D1 + D2, n = blockIdx.x, t = n / B, b = n - t * B, tid = threadIdx.x;
    float *zt = sm, *ze = sm + D, *red = sm + (size_t)(E + 1) * D;
    for (int i = tid; i < D; i += 256) zt[i] = i < D1 ? y[(size_t)n * D1 + i] : ys[(size_t)n * D2 + (i - D1)];
    for (int e = 0; e < E; ++e) {
        const size_t r = ((size_t)t * E + e) * B + b;
        for (int i = tid; i < D; i += 256) ze[(size_t)e * D + i] = i < D1 ? yp[r * D1 + i] : yps[r * D2 + (i - D1)];
    }
    __syncthreads();
    float skill = 0.0f, spread = 0.0f;
    for (int e = 0; e < E; ++e) {
        float a = 0.0f;
        for (int i = tid; i < D; i += 256) { const float d = zt[i] - ze[(size_t)e * D + i]; a += d * d; }
        skill += sqrtf(crps_block_sum(a, red));
        for (int f = e + 1; f < E; ++f) {
            float c = 0.0f;
            for (int i = tid; i < D; i += 256) { const float d = ze[(size_t)e * D + i] - ze[(size_t)f * D + i]; c += d * d; }
            spread += 2.0f * sqrtf(crps_block_sum(c, red));      // (e,f) and (f,e)
        }
    }
    if (tid == 0) { part[2 * (size_t)n] = skill; part[2 * (size_t)n + 1] = spread; }
}

__global__ __launch_bounds__(256) void crps_final_kernel(const float *__restrict__ part, int N, int E, int D, float beta, float alpha,
                                                         float *__restrict__ out)
{
    __shared__ float red[256];
    float a = 0.0f, c = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) { a += part[2 * (size_t)i]; c += part[2 * (size_t)i + 1]; }
    const float sk = crps_block_sum(a, red), sp = crps_block_sum(c, red);
    if (threadIdx.x == 0) {
        const float rs = rsqrtf((float)D), eps = (1.0f - alpha) / (float)E;
        const float mse = sk / ((float)N * (float)E) * rs;
        const float var = E > 1 ? (1.0f - eps) * (sp / (float)N) / ((float)E * (float)(E - 1)) * rs : 0.0f;
        out[0] = beta * 2.0f * mse - var;
        out[1] = mse;
        out[2] = var;
    }
}

// y (T*B, nlev*ny) / y_sfc (T*B, ny_sfc): truth; y_pred (T*E*B, nlev*ny) / y_sfc_pred: ensemble forward outputs ordered
// (time, member, column); scratch: 2*T*B floats; out: 3 device floats [CRPS, skill term, spread term]
extern "C" int csa_crps(int T, int B, int E, int D_lev, int D_sfc, const float *y, const float *y_sfc, const float *y_pred,
                        const float *y_sfc_pred, float beta, float alpha, float *scratch, float *out, void *stream)
{
    if (T <= 0 || B <= 0 || E <= 0 || D_lev <= 0 || D_sfc < 0 || !y || !y_pred || (D_sfc > 0 && (!y_sfc || !y_sfc_pred)) || !scratch || !out) {
        csa_set_error_msg("csa_crps: bad argument");
        return CSA_ERR_ARG;
    }
    const int D = D_lev + D_sfc, N = T * B;
    const size_t shm = sizeof(float) * ((size_t)(E + 1) * D + 256);
    if (shm > 64 * 1024) { csa_set_error_msg("csa_crps: ensemble x features exceed the 64 KB LDS staging (E*D too large)"); return CSA_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(crps_sample_kernel, dim3(N), dim3(256), shm, s, y, y_sfc, y_pred, y_sfc_pred, scratch, B, E, D_lev, D_sfc);
    hipLaunchKernelGGL(crps_final_kernel, dim3(1), dim3(256), 0, s, scratch, N, E, D, beta, alpha, out);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// ---- gradient of the score w.r.t. the ensemble outputs (training with the score as loss, rnn/utils.py:1213) --------------------
//   dCRPS/dz_e = 2 beta / (N E sqrt(D)) (z_e - z) / ||z - z_e||  -  2 (1 - eps) / (N E (E-1) sqrt(D)) sum_{f != e} (z_e - z_f) / ||z_e - z_f||
// (each unordered pair appears twice in the reference's cdist(y_pred, y_pred) sum; a zero distance contributes 0, as
// torch.cdist's backward does).  Same staging as the forward: one workgroup per (time, column) sample, distances by block
// reductions, then every thread writes its features of all members.  gscale = dLoss/dCRPS.
__global__ __launch_bounds__(256) void crps_backward_kernel(
    const float *__restrict__ y, const float *__restrict__ ys, const float *__restrict__ yp, const float *__restrict__ yps,
    float *__restrict__ d_yp, float *__restrict__ d_yps, int N, int B, int E, int D1, int D2, float beta, float alpha, float gscale)
{
    extern __shared__ float sm[];
    const int D = D1 + D2, n = blockIdx.x, t = n / B, b = n - t * B, tid = threadIdx.x;
    float *zt = sm, *ze = sm + D, *red = sm + (size_t)(E + 1) * D, *inv = red + 256;     // inv: E (skill) + E*E (pairs) reciprocals
    for (int i = tid; i < D; i += 256) zt[i] = i < D1 ? y[(size_t)n * D1 + i] : ys[(size_t)n * D2 + (i - D1)];
    for (int e = 0; e < E; ++e) {
        const size_t r = ((size_t)t * E + e) * B + b;
        for (int i = tid; i < D; i += 256) ze[(size_t)e * D + i] = i < D1 ? yp[r * D1 + i] : yps[r * D2 + (i - D1)];
    }
    __syncthreads();
    for (int e = 0; e < E; ++e) {
        float a = 0.0f;
        for (int i = tid; i < D; i += 256) { const float d = zt[i] - ze[(size_t)e * D + i]; a += d * d; }
        const float ds = sqrtf(crps_block_sum(a, red));
        if (tid == 0) inv[e] = ds > 0.0f ? 1.0f / ds : 0.0f;
        for (int f = e + 1; f < E; ++f) {
            float c = 0.0f;
            for (int i = tid; i < D; i += 256) { const float d = ze[(size_t)e * D + i] - ze[(size_t)f * D + i]; c += d * d; }
            const float dp = sqrtf(crps_block_sum(c, red));
            if (tid == 0) inv[E + e * E + f] = inv[E + f * E + e] = dp > 0.0f ? 1.0f / dp : 0.0f;
        }
    }
    __syncthreads();
    const float rs = rsqrtf((float)D), eps = (1.0f - alpha) / (float)E;
    const float cs = gscale * 2.0f * beta * rs / ((float)N * (float)E);
    const float cv = E > 1 ? gscale * 2.0f * (1.0f - eps) * rs / ((float)N * (float)E * (float)(E - 1)) : 0.0f;
    for (int e = 0; e < E; ++e) {
        const size_t r = ((size_t)t * E + e) * B + b;
        for (int i = tid; i < D; i += 256) {
            const float x = ze[(size_t)e * D + i];
            float g = cs * (x - zt[i]) * inv[e];
            for (int f = 0; f < E; ++f)
                if (f != e) g -= cv * (x - ze[(size_t)f * D + i]) * inv[E + e * E + f];
            if (i < D1) d_yp[r * D1 + i] = g; else d_yps[r * D2 + (i - D1)] = g;
        }
    }
}

extern "C" int csa_crps_backward(int T, int B, int E, int D_lev, int D_sfc, const float *y, const float *y_sfc, const float *y_pred,
                                 const float *y_sfc_pred, float beta, float alpha, float gscale, float *d_y_pred,
                                 float *d_y_sfc_pred, void *stream)
{
    if (T <= 0 || B <= 0 || E <= 0 || D_lev <= 0 || D_sfc < 0 || !y || !y_pred || !d_y_pred ||
        (D_sfc > 0 && (!y_sfc || !y_sfc_pred || !d_y_sfc_pred))) {
        csa_set_error_msg("csa_crps_backward: bad argument");
        return CSA_ERR_ARG;
    }
    const int D = D_lev + D_sfc, N = T * B;
    const size_t shm = sizeof(float) * ((size_t)(E + 1) * D + 256 + E + (size_t)E * E);
    if (shm > 64 * 1024) { csa_set_error_msg("csa_crps_backward: ensemble x features exceed the 64 KB LDS staging (E*D too large)"); return CSA_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(crps_backward_kernel, dim3(N), dim3(256), shm, (hipStream_t)stream, y, y_sfc, y_pred, y_sfc_pred, d_y_pred,
                       d_y_sfc_pred, N, B, E, D_lev, D_sfc, beta, alpha, gscale);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// rnn/metrics.py:509-533 compute_spread_skill_ratio (logged next to the ensemble score in the stochastic training loop,
// rnn/utils.py:1214-1215) and rnn/metrics.py:628-699 CRPS_l1 (the two-member L1 form: mean|z_e - z| - 0.5 mean|z_0 - z_1|).
// One pass over the ensemble outputs: a thread owns one (sample, feature) cell, walks its E members (consecutive lanes ->
// consecutive features: coalesced) and produces the unbiased member variance (shifted by member 0, so no cancellation),
// the squared error of the member mean, the summed |z_e - z| and |z_0 - z_1|; float64 workgroup partials in a fixed
// tree order, a single-workgroup second pass adds them in a fixed order (deterministic).
#define SS_BLOCKS 1024

__global__ __launch_bounds__(256) void spread_skill_partial_kernel(
    const float *__restrict__ y, const float *__restrict__ ys, const float *__restrict__ yp, const float *__restrict__ yps,
    double *__restrict__ part, int T, int B, int E, int D1, int D2)
{
    __shared__ double red[4][256];
    const int D = D1 + D2, tid = threadIdx.x;
    const long total = (long)T * B * D;
    double a_var = 0.0, a_se = 0.0, a_l1 = 0.0, a_d01 = 0.0;
    for (long c = (long)blockIdx.x * 256 + tid; c < total; c += (long)gridDim.x * 256) {
        const long n = c / D;
        const int d = (int)(c - n * D), t = (int)(n / B), b = (int)(n - (long)t * B);
        const float yt = d < D1 ? y[n * D1 + d] : ys[n * D2 + (d - D1)];
        float x0 = 0.0f, x1 = 0.0f, s1 = 0.0f, s2 = 0.0f, l1 = 0.0f;
        for (int e = 0; e < E; ++e) {
            const long r = ((long)t * E + e) * B + b;
            const float x = d < D1 ? yp[r * D1 + d] : yps[r * D2 + (d - D1)];
            if (e == 0) x0 = x;
            if (e == 1) x1 = x;
            const float u = x - x0;
            s1 += u;
            s2 += u * u;
            l1 += fabsf(x - yt);
        }
        const float mean = x0 + s1 / (float)E;
        a_var += E > 1 ? (double)((s2 - s1 * s1 / (float)E) / (float)(E - 1)) : 0.0;
        a_se += (double)(mean - yt) * (double)(mean - yt);
        a_l1 += l1;
        a_d01 += E > 1 ? fabsf(x0 - x1) : 0.0f;
    }
    red[0][tid] = a_var; red[1][tid] = a_se; red[2][tid] = a_l1; red[3][tid] = a_d01;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + s];
        __syncthreads();
    }
    if (tid < 4) part[4 * (size_t)blockIdx.x + tid] = red[tid][0];
}

__global__ __launch_bounds__(256) void spread_skill_final_kernel(const double *__restrict__ part, int nblk, double cells, int E,
                                                                 float *__restrict__ out)
{
    __shared__ double red[4][256];
    const int tid = threadIdx.x;
    double a[4] = {0, 0, 0, 0};
    for (int i = tid; i < nblk; i += 256) for (int k = 0; k < 4; ++k) a[k] += part[4 * (size_t)i + k];
    for (int k = 0; k < 4; ++k) red[k][tid] = a[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = (float)(sqrt(red[0][0] / cells) * sqrt((double)(E + 1) / (double)E));   // spread, with the sqrt((M+1)/M) correction
        out[1] = (float)sqrt(red[1][0] / cells);                                         // RMSE of the member mean
        const double skill = red[2][0] / (cells * E), d01 = red[3][0] / cells;
        out[2] = (float)(skill - 0.5 * d01);                                             // CRPS_l1 (meaningful for E = 2, as in the reference)
        out[3] = (float)skill;
    }
}

// same tensor arguments as csa_crps; scratch: 4 * 1024 doubles (32 KB); out: 4 device floats
// [spread, rmse (compute_spread_skill_ratio), CRPS_l1, its skill term]
extern "C" int csa_spread_skill(int T, int B, int E, int D_lev, int D_sfc, const float *y, const float *y_sfc,
                                const float *y_pred, const float *y_sfc_pred, void *scratch, float *out, void *stream)
{
    if (T <= 0 || B <= 0 || E <= 0 || D_lev <= 0 || D_sfc < 0 || !y || !y_pred || (D_sfc > 0 && (!y_sfc || !y_sfc_pred)) || !scratch || !out) {
        csa_set_error_msg("csa_spread_skill: bad argument");
        return CSA_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const double cells = (double)T * B * (D_lev + D_sfc);
    const long nb = ((long)cells + 255) / 256;
    const int nblk = (int)(nb < SS_BLOCKS ? nb : SS_BLOCKS);
    hipLaunchKernelGGL(spread_skill_partial_kernel, dim3(nblk), dim3(256), 0, s, y, y_sfc, y_pred, y_sfc_pred, (double *)scratch,
                       T, B, E, D_lev, D_sfc);
    hipLaunchKernelGGL(spread_skill_final_kernel, dim3(1), dim3(256), 0, s, (const double *)scratch, nblk, cells, E, out);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
