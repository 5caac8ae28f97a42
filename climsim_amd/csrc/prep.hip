// prep.hip -- fused wrapper pre-processing + pressure feature + mlp_initial + memory concat
// + surface/TOA initial-state MLPs.  One workgroup per grid column.
//
// Reference semantics (paths relative to the reference root):
//   preprocessing      rnn/utils.py:182-217, rnn/save_wrapper_mem.py:411-457
//   LayerPressure      rnn/layers.py:117-121 (sqrt(p)/314), call site rnn/models/models.py:446-453
//   mlp_initial+tanh   rnn/models/models.py:457-458
//   memory concat+flip rnn/models/models.py:461,478 (current); legacy artefacts concatenate the
//                      memory AFTER the flip, i.e. it is indexed by sequence position
//   mlp_surface1/2     rnn/models/models.py:485-491 (legacy: c0 = tanh(.))
//   mlp_toa1/2         rnn/models/models.py:503-506
//
// HBM-bound elementwise kernel: reads 3.6 KB + 76 B (+3.8 KB memory) per column, writes the
// rnn1 input rows X1 (L,B,nh1+nh_mem) in SEQUENCE order (t = 0 is the surface level).
#include "common.h"

#define PREP_THREADS 128
#define PREP_MAX_NXP 32   // nx+1 upper bound held in registers
#define PREP_LSPLIT 4     // workgroups per column (each owns a contiguous slice of levels)

// tanh for the mlp_initial activation: (1 - t)/(1 + t), t = exp(-2x); absolute error ~6e-8.
__device__ __forceinline__ float prep_tanh(float x)
{
    const float t = fminf(__builtin_amdgcn_exp2f(-2.88539008177792681f * x), 1e30f);
    return (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
}

#include "rh_to_q.h"

// NXP = nx+1 padded to a multiple of 4 (16 for the v4 inputs): the level rows sit in LDS with that
// stride so the mlp_initial dot product reads them as float4 broadcasts with no predication.
template <int NXP>
__global__ __launch_bounds__(PREP_THREADS) void prep_kernel(
    DevModel m, int B, int normalised,
    const float *__restrict__ x_main, const float *__restrict__ x_sfc,
    const float *__restrict__ mem_in, float *__restrict__ X1, float *__restrict__ hc0,
    float *__restrict__ X16out, float *__restrict__ xs_out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = m.cfg.nlev, nx = m.cfg.nx, nxp = nx + 1, nxs = m.cfg.nx_sfc;
    const int nh1 = m.cfg.nh1, nh2 = m.cfg.nh2, nm = m.cfg.nh_mem, nin1 = nh1 + nm;
    const int nhm = nh1 > nh2 ? nh1 : nh2;
    const int b = blockIdx.x, tid = threadIdx.x;
    // this workgroup's slice of levels [l0, l1)
    const int lper = (L + PREP_LSPLIT - 1) / PREP_LSPLIT;
    const int l0 = blockIdx.y * lper, l1 = min(L, l0 + lper), nl = l1 - l0;
    float *xl = smem;                 // (lper, NXP), zero padded beyond nxp
    float *xs = smem + lper * NXP;    // (nxs)
    if (nl <= 0) return;
    // the mlp_initial column of this thread's first unit is fetched now, so that its L2 round trip overlaps the input
    // loads and the two barriers below instead of following them
    float w0[NXP];
    float bj0 = 0.0f;
    if (tid < nh1) {
#pragma unroll
        for (int v = 0; v < NXP; ++v) w0[v] = v < nxp ? m.init_wt[v * nh1 + tid] : 0.0f;
        bj0 = m.init_b[tid];
    }

    // ---- surface inputs -----------------------------------------------------------------
    for (int v = tid; v < nxs; v += PREP_THREADS) {
        float x = x_sfc[(size_t)b * nxs + v];
        if (!normalised) {
            if (m.cfg.snowhice_fix && x >= 1e10f) x = -1.0f;
            x = (x - m.xmean_sca[v]) / m.xdiv_sca[v];
        }
        xs[v] = x;
    }
    // ---- level inputs of the slice ------------------------------------------------------------
    for (int idx = tid; idx < nl * nx; idx += PREP_THREADS) {
        const int ll = idx / nx, v = idx - ll * nx, l = l0 + ll, gi = l * nx + v;
        const int qmode = normalised ? 0 : m.cfg.q_input_mode, nxr = nx - (qmode == 1);
        const float *xrow = x_main + ((size_t)b * L + l) * nxr;
        float x = v < nxr ? xrow[v] : 0.0f;
        if ((qmode == 1 && v == nxr) || (qmode == 2 && v == 1)) {
            const float pres = m.hyam[l] * 100000.0f + x_sfc[(size_t)b * nxs] * m.hybm[l];   // RAW surface pressure
            x = prep_rh_to_q(xrow[1], xrow[0], pres);
        }
        if (!normalised) {
            if (m.cfg.v5_input) {      // rnn/utils.py:186-198: total cloud water with its own rate, liquid fraction from T
                if (v == 2) {
                    x = xrow[2] + xrow[3];
                    if (m.cfg.qinput_prune && l < 15) x = 0.0f;      // pruned BEFORE the transform (:188-190)
                    x = 1.0f - expf(-x * m.lbd_qn[l]);
                }
                if (v == 3) x = fminf(fmaxf((xrow[0] - 253.16f) * 0.05f, 0.0f), 1.0f);   // models.py:260-266
            } else {
                if (v == 2) x = 1.0f - expf(-x * m.lbd_qc[l]);
                if (v == 3) x = 1.0f - expf(-x * m.lbd_qi[l]);
            }
            x = (x - m.xmean_lev[gi]) / m.xdiv_lev[gi];
            if (!m.cfg.v5_input && m.cfg.qinput_prune && v == 2 && l < 15) x = 0.0f;
            if (m.cfg.rh_prune && v == 1 && !isnan(x)) x = fminf(fmaxf(x, 0.0f), 1.2f);
            if (isnan(x)) x = 0.0f;
            if (m.cfg.scrub_inf && isinf(x)) x = 0.0f;
        }
        xl[ll * NXP + v] = x;
    }
    if (NXP > nxp)
        for (int idx = tid; idx < nl * (NXP - nxp); idx += PREP_THREADS)
            xl[(idx / (NXP - nxp)) * NXP + nxp + idx % (NXP - nxp)] = 0.0f;
    __syncthreads();
    {
        const float sp = xs[0] * m.xdiv_sca[0] + m.xmean_sca[0];
        for (int ll = tid; ll < nl; ll += PREP_THREADS) {
            const int l = l0 + ll;
            const float pres = m.hyam[l] * 100000.0f + sp * m.hybm[l];
            xl[ll * NXP + nx] = sqrtf(pres) / 314.0f;
        }
    }
    __syncthreads();
    if (X16out) {   // training: keep what the backward of mlp_initial / the surface MLPs needs
        for (int idx = tid; idx < nl * nxp; idx += PREP_THREADS)
            X16out[((size_t)b * L + l0) * nxp + idx] = xl[(idx / nxp) * NXP + idx % nxp];
        if (blockIdx.y == 0)
            for (int v = tid; v < nxs; v += PREP_THREADS) xs_out[(size_t)b * nxs + v] = xs[v];
    }

    // ---- initial states (first slice only) ---------------------------------------------------
    if (blockIdx.y == 0) {
        for (int j = tid; j < nh1; j += PREP_THREADS) {
            // chunks of 8 with a clamped weight index and a select on the input: compile-time trip counts, so the eight
            // (sixteen) weight loads of a chunk are issued together instead of one s_waitcnt per element
            float a = m.s1_b[j], c = m.cfg.use_lstm ? m.s2_b[j] : 0.0f;
            for (int v0 = 0; v0 < nxs; v0 += 8) {
                float w1[8], w2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int v = min(v0 + u, nxs - 1);
                    w1[u] = m.s1_wt[v * nh1 + j];
                    w2[u] = m.cfg.use_lstm ? m.s2_wt[v * nh1 + j] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float x = v0 + u < nxs ? xs[min(v0 + u, nxs - 1)] : 0.0f;
                    a += w1[u] * x;
                    c += w2[u] * x;
                }
            }
            hc0[((size_t)0 * B + b) * nhm + j] = tanhf(a);
            if (m.cfg.use_lstm) hc0[((size_t)1 * B + b) * nhm + j] = m.cfg.legacy ? tanhf(c) : c;
        }
        if (!m.cfg.legacy) {
            const float t0 = xs[1], t1 = xs[6];
            for (int j = tid; j < nh2; j += PREP_THREADS) {
                hc0[((size_t)2 * B + b) * nhm + j] = m.toa1_b[j] + m.toa1_wt[j] * t0 + m.toa1_wt[nh2 + j] * t1;
                if (m.cfg.use_lstm)
                    hc0[((size_t)3 * B + b) * nhm + j] = m.toa2_b[j] + m.toa2_wt[j] * t0 + m.toa2_wt[nh2 + j] * t1;
            }
        }
    }

    // ---- mlp_initial + tanh, written in sequence order (t = L-1-l) ------------------------------
    for (int j = tid; j < nh1; j += PREP_THREADS) {
        float w[NXP];
        float bj = bj0;
#pragma unroll
        for (int v = 0; v < NXP; ++v) w[v] = w0[v];
        if (j != tid) {
#pragma unroll
            for (int v = 0; v < NXP; ++v) w[v] = v < nxp ? m.init_wt[v * nh1 + j] : 0.0f;
            bj = m.init_b[j];
        }
        for (int ll = 0; ll < nl; ++ll) {
            const f32x4 *xr = (const f32x4 *)(xl + ll * NXP);
            float a0 = bj, a1 = 0.0f;
#pragma unroll
            for (int q = 0; q < NXP / 4; ++q) {
                const f32x4 xv = xr[q];
                a0 += w[4 * q] * xv.x; a1 += w[4 * q + 1] * xv.y;
                a0 += w[4 * q + 2] * xv.z; a1 += w[4 * q + 3] * xv.w;
            }
            const int t = m.cfg.add_stochastic_layer ? l0 + ll : L - 1 - (l0 + ll);   // rnn0 runs downward
            X1[((size_t)t * B + b) * nin1 + j] = prep_tanh(a0 + a1);
        }
    }
    // ---- memory concat -------------------------------------------------------------------------
    for (int idx = tid; idx < nl * nm; idx += PREP_THREADS) {
        const int ll = idx / nm, k = idx - ll * nm, t = m.cfg.add_stochastic_layer ? l0 + ll : L - 1 - (l0 + ll);
        const float v = m.cfg.legacy ? mem_in[((size_t)b * L + t) * nm + k]
                                     : mem_in[((size_t)(l0 + ll) * (m.mem_B > 0 ? m.mem_B : B) + m.mem_off + b) * nm + k];
        X1[((size_t)t * B + b) * nin1 + nh1 + k] = v;
    }
}

static int launch_prep_impl(const DevModel &m, int B, int normalised, const float *x_main, const float *x_sfc,
                            const float *mem_in, float *X1, float *hc0, float *X16, float *xs_n, hipStream_t s)
{
    const int lper = (m.cfg.nlev + PREP_LSPLIT - 1) / PREP_LSPLIT;
    const int nxp = m.cfg.nx + 1, NXP = nxp <= 16 ? 16 : 32;
    const size_t shm = sizeof(float) * ((size_t)lper * NXP + m.cfg.nx_sfc + 4);
    const dim3 grid(B, PREP_LSPLIT), block(PREP_THREADS);
    if (NXP == 16)
        hipLaunchKernelGGL(prep_kernel<16>, grid, block, shm, s, m, B, normalised, x_main, x_sfc, mem_in, X1, hc0, X16, xs_n);
    else
        hipLaunchKernelGGL(prep_kernel<32>, grid, block, shm, s, m, B, normalised, x_main, x_sfc, mem_in, X1, hc0, X16, xs_n);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

int launch_prep(const DevModel &m, int B, int normalised, const float *x_main, const float *x_sfc,
                const float *mem_in, const float *hx2, const float *cx2,
                float *X1, float *hc0, hipStream_t s)
{
    (void)hx2; (void)cx2;
    if (m.cfg.nx + 1 > PREP_MAX_NXP) {
        csa_set_error_msg("prep: nx+1 exceeds PREP_MAX_NXP");
        return CSA_ERR_UNSUPPORTED;
    }
    return launch_prep_impl(m, B, normalised, x_main, x_sfc, mem_in, X1, hc0, nullptr, nullptr, s);
}

int launch_prep_train(const DevModel &m, int B, int normalised, const float *x_main, const float *x_sfc,
                      const float *mem_in, float *X1, float *hc0, float *X16, float *xs_n, hipStream_t s)
{
    if (m.cfg.nx + 1 > PREP_MAX_NXP) {
        csa_set_error_msg("prep: nx+1 exceeds PREP_MAX_NXP");
        return CSA_ERR_UNSUPPORTED;
    }
    return launch_prep_impl(m, B, normalised, x_main, x_sfc, mem_in, X1, hc0, X16, xs_n, s);
}
