// gen.hip -- the training-time twin of the wrapper pre-processing plus the target construction, as ONE device-side
// pass over a loaded chunk: generator_xy.__getitem__ (rnn/utils.py:2238-2371) and its numba kernels (:1795-1868).
// The reference does this on host cores inside DataLoader workers and names it the training bottleneck
// (rnn/train_rnn_rollout_torchscript_hydra.py:592-593); here the raw chunk is copied to the GPU once and a single
// HBM-bound kernel (one thread per (sample, level)) emits all seven tensors the trainer consumes.
//   inputs : reverse reference scaling -> drop the 5 past-state scalars -> snow/ice sentinel -> RH clip -> RH -> q
//            (float64 Horner, as numpy's polyval with float64 coefficients does) -> raw copy -> cloud transform
//            (exp / sqrt / v4->v5 qn + liquid fraction) -> q-input prune -> new scaling -> q >= 0 -> NaN -> 0
//   targets: reverse reference scaling -> raw copies -> mp_mode 1 (qn = qliq + qice), -1 (qn + liquid fraction),
//            -2 (total water + cloud fraction^(1/4) + liquid fraction) -> * yscale -> output prune
#include "common.h"
#include "rh_to_q.h"
#include <vector>

struct GenDev {
    csa_gen_config c;
    int nx_out, nxs_out, ny_out;
    const float *xmean_lev, *xdiv_lev, *xmean_sca, *xdiv_sca, *yscale_lev, *yscale_sca;
    const float *lbd_qc, *lbd_qi, *lbd_qn, *hyam, *hybm;
    const float *xref_mean, *xref_div, *xsref_mean, *xsref_div, *yref_lev, *yref_sca;
};

struct csa_generator {
    GenDev d;
    std::vector<void *> owned;
};

#pragma clang fp contract(off)   // numpy evaluates every product and sum separately
__device__ __forceinline__ double gen_polyval9(const double *a, double x)
{
    double o = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) o = o * x + a[i];     // separate rounding of * and + (no contraction, below)
    return o;
}

// relative_to_specific_humidity_climsim (rnn/utils.py:647-702): numpy promotes to float64 through np.polyval's
// float64 coefficient arrays; the result is rounded to float32 once (np.float32(...), :2190)
__device__ float gen_rh_to_q(float rh, float T, float p)
{
    const double a_liq[9] = {-0.976195544e-15, -0.952447341e-13, 0.640689451e-10, 0.206739458e-7, 0.302950461e-5,
                             0.264847430e-3, 0.142986287e-1, 0.443987641, 6.11239921};
    const double a_ice[9] = {0.252751365e-14, 0.146898966e-11, 0.385852041e-9, 0.602588177e-7, 0.615021634e-5,
                             0.420895665e-3, 0.188439774e-1, 0.503160820, 6.11147274};
    const float T0f = 273.16f, T00f = 253.16f;
    // (temp - T00) / (T0 - T00): float32 array with python scalars -> float32
    float omega = (T - T00f) / (T0f - T00f);
    omega = fmaxf(0.0f, fminf(1.0f, omega));
    const float dTf = T - T0f;                                       // float32 array - python float
    const double eliq = 100.0 * gen_polyval9(a_liq, (double)fmaxf(-80.0f, dTf));
    double eice;
    if (T > 273.15f) eice = eliq;
    else if (T > 185.0f) eice = 100.0 * gen_polyval9(a_ice, (double)dTf);
    else {
        const double m = (double)fmaxf(-100.0f, dTf);
        eice = 100.0 * (0.00763685 + m * (0.000151069 + m * 7.48215e-07));
    }
    const double esat = (double)omega * eliq + (double)(1.0f - omega) * eice;
    const double qvs = (287.0 * esat) / (double)(461.0f * p);         // Rv*pressure: int * float32 array -> float32
    return (float)((double)rh * qvs);
}

// One workgroup = GEN_T consecutive (sample, level) rows.  The row-major tensors are moved between HBM and LDS with
// fully coalesced block copies (a thread-per-row access touches 64 different cache lines per wave instruction), and
// everything that is elementwise -- reversing the reference scaling, the new scaling with its per-(level, variable)
// tables and IEEE divisions, the NaN scrub, the target scaling -- is done by those copy loops, one independent
// iteration per element.  Only the operations that couple the variables of a row (RH clip and RH -> q, the cloud
// transforms, the microphysics targets) run thread-per-row, on the row in LDS (odd stride 17 / 33: conflict-free).
#define GEN_T 128
#define GEN_SY 7
__global__ __launch_bounds__(GEN_T) void gen_batch_kernel(
    GenDev g, int N, const float *__restrict__ x_lev, const float *__restrict__ x_sfc, const float *__restrict__ y_lev,
    const float *__restrict__ y_sfc, float *__restrict__ xo, float *__restrict__ xso, float *__restrict__ yo,
    float *__restrict__ yso, float *__restrict__ xd, float *__restrict__ yd, float *__restrict__ ysd)
{
    extern __shared__ float gsm[];             // sx[GEN_T][SX] | sy[GEN_T][7] | level of each row
    const int GEN_SX = g.nx_out <= 16 ? 17 : 33;
    float *sx = gsm, *sy = sx + GEN_T * GEN_SX;
    int *sl = (int *)(sy + GEN_T * GEN_SY);
    const csa_gen_config &c = g.c;
    const int L = c.nlev, tid = threadIdx.x;
    const long total = (long)N * L, i0 = (long)blockIdx.x * GEN_T, i = i0 + tid;
    const int rows = (int)(total - i0 < GEN_T ? total - i0 : GEN_T);
    const bool valid = tid < rows;
    const int n = valid ? (int)(i / L) : 0, l = valid ? (int)(i - (long)n * L) : 0;
    const int nxi = c.nx_in, nxo = g.nx_out, ny = g.ny_out;
    sl[tid] = l;
    __syncthreads();
    // (e + 0.5) / width is >= 0.5 / 33 away from an integer, so the float quotient gives the exact row index
    // ---- coalesced block loads, reference scaling reversed on the way in ---------------------------------------------
    {
        const float *xb = x_lev + (size_t)i0 * nxi, *yb = y_lev + (size_t)i0 * 6;
        const float invx = 1.0f / (float)nxi;
        for (int e = tid; e < rows * nxi; e += GEN_T) {
            const int r = (int)(((float)e + 0.5f) * invx), v = e - r * nxi;
            float t = xb[e];
            if (c.reverse_input_norm) { const int k = sl[r] * nxi + v; t = add_nofma(mul_nofma(t, g.xref_div[k]), g.xref_mean[k]); }
            sx[r * GEN_SX + v] = t;
        }
#pragma unroll 4
        for (int e = tid; e < rows * 6; e += GEN_T) {
            const int r = e / 6, v = e - r * 6;
            float t = yb[e];
            if (c.reverse_output_norm) t = t / g.yref_lev[sl[r] * 6 + v];
            sy[r * GEN_SY + v] = t;
        }
    }
    __syncthreads();

    // ---- scalars: every thread needs the raw surface pressure; level 0 writes the outputs ------------------
    const float *xs = x_sfc + (size_t)n * c.nx_sfc_in;
    auto sfc_raw = [&](int vin) {
        float v = xs[vin];
        if (c.reverse_input_norm) v = add_nofma(mul_nofma(v, g.xsref_div[vin]), g.xsref_mean[vin]);
        if (c.snowhice_fix && v > 1.0e10f) v = -1.0f;
        return v;
    };
    if (valid && l == 0) {
        for (int vo = 0; vo < g.nxs_out; ++vo) {
            const int vin = (c.remove_past_sfc_inputs && vo >= 17) ? vo + 5 : vo;
            float v = sfc_raw(vin);
            if (c.apply_new_input_scaling) v = (v - g.xmean_sca[vo]) / g.xdiv_sca[vo];
            xso[(size_t)n * g.nxs_out + vo] = v;
        }
        for (int v = 0; v < c.ny_sfc; ++v) {
            float t = y_sfc[(size_t)n * c.ny_sfc + v];
            if (c.reverse_output_norm) t = t / g.yref_sca[v];
            ysd[(size_t)n * c.ny_sfc + v] = t;
            yso[(size_t)n * c.ny_sfc + v] = t * g.yscale_sca[v];
        }
    }

    // ---- level inputs: row-coupled part 1 (RH clip, RH -> q) -----------------------------------------------------
    float *x = sx + tid * GEN_SX, *y = sy + tid * GEN_SY;
    const int qcol = c.q_mode == 0 ? -1 : (c.q_mode == 1 ? nxi : 1);
    float T_b = 0.f, ql_b = 0.f, qi_b = 0.f, qlast_b = 0.f;
    if (valid) {
        if (c.rh_prune) x[1] = fminf(fmaxf(x[1], 0.0f), 1.2f);       // np.clip: NaN propagates
        if (c.q_mode != 0) {
            const float sp = sfc_raw(0);
            const float pres = add_nofma(mul_nofma(sp, g.hybm[l]), mul_nofma(100000.0f, g.hyam[l]));
            x[qcol] = gen_rh_to_q(x[1], x[0], pres);
        }
        T_b = x[0]; ql_b = x[2]; qi_b = x[3]; qlast_b = x[nxo - 1];
    }
    __syncthreads();
    {   // x_lev_b_denorm (raw, q included): plain block copy
        float *d = xd + (size_t)i0 * nxo;
        const float inv = 1.0f / (float)nxo;
#pragma unroll 4
        for (int e = tid; e < rows * nxo; e += GEN_T) { const int r = (int)(((float)e + 0.5f) * inv); d[e] = sx[r * GEN_SX + (e - r * nxo)]; }
    }
    __syncthreads();
    // ---- row-coupled part 2: cloud transforms ------------------------------------------------------------------------
    if (valid) {
        if (c.v4_to_v5_inputs) {
            float lf = mul_nofma(x[0] - 253.16f, 0.05f);
            lf = fminf(fmaxf(lf, 0.0f), 1.0f);
            float qn = x[2] + x[3];
            if (c.qinput_prune && l < 15) qn = 0.0f;
            if (c.cld_inp_transformation == 1) qn = 1.0f - expf(-qn * g.lbd_qn[l]);
            else if (c.cld_inp_transformation == 2) qn = sqrtf(sqrtf(qn));
            x[2] = qn; x[3] = lf;
        } else {
            if (c.cld_inp_transformation == 1) {
                x[2] = 1.0f - expf(-x[2] * g.lbd_qc[l]);
                x[3] = 1.0f - expf(-x[3] * g.lbd_qi[l]);
            } else if (c.cld_inp_transformation == 2) {
                x[2] = sqrtf(sqrtf(x[2]));
                x[3] = sqrtf(sqrtf(x[3]));
            }
            if (c.qinput_prune && l < 15) x[2] = 0.0f;
        }
    }
    __syncthreads();
    {   // new scaling, q >= 0, NaN -> 0: per element, straight to global memory
        float *d = xo + (size_t)i0 * nxo;
        const float inv = 1.0f / (float)nxo;
#pragma unroll 4
        for (int e = tid; e < rows * nxo; e += GEN_T) {
            const int r = (int)(((float)e + 0.5f) * inv), v = e - r * nxo;
            float t = sx[r * GEN_SX + v];
            if (c.apply_new_input_scaling) {
                const int k = sl[r] * nxo + v;
                t = (t - g.xmean_lev[k]) / g.xdiv_lev[k];
                if (v == qcol && t < 0.0f) t = 0.0f;
            }
            if (isnan(t)) t = 0.0f;
            d[e] = t;
        }
    }
    // ---- level targets: raw copy, then the row-coupled microphysics targets, then scaling -------------------------
    {
        float *d = yd + (size_t)i0 * 6;
#pragma unroll 4
        for (int e = tid; e < rows * 6; e += GEN_T) { const int r = e / 6; d[e] = sy[r * GEN_SY + (e - r * 6)]; }
    }
    __syncthreads();
    if (valid) {
        const float y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3], y4 = y[4], y5 = y[5];
        if (c.mp_mode > 0) {                 // hu_mp_constraint: qn = qliq + qice replaces the two cloud tendencies
            y[2] = y2 + y3; y[3] = y4; y[4] = y5;
        } else if (c.mp_mode < 0) {          // pred_liq_frac (rnn/utils.py:2295-2343), float32 numpy arithmetic, unfused
            const float qn_b = add_nofma(ql_b, qi_b);
            const float dqn = add_nofma(y2, y3);
            float qn_new = add_nofma(qn_b, mul_nofma(dqn, 1200.0f));
            if (qn_new < 0.0f) qn_new = 0.0f;
            const float ql_new = add_nofma(ql_b, mul_nofma(y2, 1200.0f));
            const float T_new = add_nofma(T_b, mul_nofma(y0, 1200.0f));
            float lf = (T_new - 253.16f) / 20.0f;
            if (lf < 0.0f) lf = 0.0f;
            if (lf > 1.0f) lf = 1.0f;
            if (qn_new > 1e-20f && dqn > 1e-20f) lf = ql_new / qn_new;
            if (lf < 0.0f) lf = 0.0f;
            if (lf > 1.0f) lf = 1.0f;
            y[2] = dqn; y[3] = lf;
            if (c.mp_mode == -2) {
                const float dqv = y1;
                float qv_new = add_nofma(qlast_b, mul_nofma(dqv, 1200.0f));
                if (qv_new < 0.0f) qv_new = 0.0f;
                const float qtot_new = add_nofma(qv_new, qn_new);
                float tcf = qtot_new > 0.0f ? qn_new / qtot_new : 0.0f;
                tcf = sqrtf(sqrtf(tcf));
                y[1] = add_nofma(dqv, dqn);
                y[2] = tcf;
            }
        }
    }
    __syncthreads();
    {
        float *d = yo + (size_t)i0 * ny;
        const float inv = 1.0f / (float)ny;
#pragma unroll 4
        for (int e = tid; e < rows * ny; e += GEN_T) {
            const int r = (int)(((float)e + 0.5f) * inv), v = e - r * ny, lr = sl[r];
            float t = sy[r * GEN_SY + v] * g.yscale_lev[lr * ny + v];
            if (c.output_prune && lr < 12 && v >= 1) t = 0.0f;
            d[e] = t;
        }
    }
}

static const float *gen_up(csa_generator *h, const float *src, size_t n, int &rc)
{
    if (!src) return nullptr;
    void *p = nullptr;
    if (hipMalloc(&p, sizeof(float) * (n ? n : 1)) != hipSuccess) { rc = CSA_ERR_NOMEM; return nullptr; }
    h->owned.push_back(p);
    if (hipMemcpy(p, src, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
    return (const float *)p;
}

extern "C" int csa_gen_create(const csa_gen_config *cfg, const csa_gen_coeffs *k, csa_generator **out)
{
    if (!cfg || !k || !out) { csa_set_error_msg("csa_gen_create: null argument"); return CSA_ERR_ARG; }
    const csa_gen_config &c = *cfg;
    const int nx_out = c.nx_in + (c.q_mode == 1), nxs_out = c.nx_sfc_in - (c.remove_past_sfc_inputs ? 5 : 0);
    const int ny_out = c.mp_mode > 0 ? 5 : 6;
    if (c.nlev <= 0 || c.nx_in < 4 || nx_out > 32 || c.nx_sfc_in <= (c.remove_past_sfc_inputs ? 22 : 0) || c.ny_sfc <= 0 ||
        c.q_mode < 0 || c.q_mode > 2 || c.cld_inp_transformation < 0 || c.cld_inp_transformation > 2 || c.mp_mode < -2 || c.mp_mode > 1) {
        csa_set_error_msg("csa_gen_create: bad configuration");
        return CSA_ERR_ARG;
    }
    if ((c.apply_new_input_scaling && !(k->xmean_lev && k->xdiv_lev && k->xmean_sca && k->xdiv_sca)) || !k->yscale_lev || !k->yscale_sca ||
        (c.q_mode && !(k->hyam && k->hybm)) || (c.reverse_input_norm && !(k->xref_mean && k->xref_div && k->xsref_mean && k->xsref_div)) ||
        (c.reverse_output_norm && !(k->yref_lev && k->yref_sca)) ||
        (c.cld_inp_transformation == 1 && (c.v4_to_v5_inputs ? !k->lbd_qn : !(k->lbd_qc && k->lbd_qi)))) {
        csa_set_error_msg("csa_gen_create: a coefficient array required by the configuration is missing");
        return CSA_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { csa_set_error_msg("csa_gen_create: no HIP device"); return CSA_ERR_HIP; }
    csa_generator *h = new csa_generator();
    int rc = CSA_OK;
    GenDev &d = h->d;
    d.c = c; d.nx_out = nx_out; d.nxs_out = nxs_out; d.ny_out = ny_out;
    const size_t L = c.nlev;
    d.xmean_lev = gen_up(h, k->xmean_lev, L * nx_out, rc); d.xdiv_lev = gen_up(h, k->xdiv_lev, L * nx_out, rc);
    d.xmean_sca = gen_up(h, k->xmean_sca, nxs_out, rc); d.xdiv_sca = gen_up(h, k->xdiv_sca, nxs_out, rc);
    d.yscale_lev = gen_up(h, k->yscale_lev, L * ny_out, rc); d.yscale_sca = gen_up(h, k->yscale_sca, c.ny_sfc, rc);
    d.lbd_qc = gen_up(h, k->lbd_qc, L, rc); d.lbd_qi = gen_up(h, k->lbd_qi, L, rc); d.lbd_qn = gen_up(h, k->lbd_qn, L, rc);
    d.hyam = gen_up(h, k->hyam, L, rc); d.hybm = gen_up(h, k->hybm, L, rc);
    d.xref_mean = gen_up(h, k->xref_mean, L * c.nx_in, rc); d.xref_div = gen_up(h, k->xref_div, L * c.nx_in, rc);
    d.xsref_mean = gen_up(h, k->xsref_mean, c.nx_sfc_in, rc); d.xsref_div = gen_up(h, k->xsref_div, c.nx_sfc_in, rc);
    d.yref_lev = gen_up(h, k->yref_lev, L * 6, rc); d.yref_sca = gen_up(h, k->yref_sca, c.ny_sfc, rc);
    if (rc) { for (void *p : h->owned) (void)hipFree(p); delete h; csa_set_error_msg("csa_gen_create: upload failed"); return rc; }
    *out = h;
    return CSA_OK;
}

extern "C" int csa_gen_destroy(csa_generator *h)
{
    if (!h) return CSA_ERR_ARG;
    for (void *p : h->owned) (void)hipFree(p);
    delete h;
    return CSA_OK;
}

extern "C" int csa_gen_dims(const csa_generator *h, int *nx_out, int *nx_sfc_out, int *ny_out)
{
    if (!h) return CSA_ERR_ARG;
    if (nx_out) *nx_out = h->d.nx_out;
    if (nx_sfc_out) *nx_sfc_out = h->d.nxs_out;
    if (ny_out) *ny_out = h->d.ny_out;
    return CSA_OK;
}

extern "C" int csa_gen_batch(csa_generator *h, int N, const float *x_lev, const float *x_sfc, const float *y_lev, const float *y_sfc,
                             float *x_lev_n, float *x_sfc_n, float *y_lev_n, float *y_sfc_n, float *x_lev_denorm,
                             float *y_lev_denorm, float *y_sfc_denorm, void *stream)
{
    if (!h || N <= 0 || !x_lev || !x_sfc || !y_lev || !y_sfc || !x_lev_n || !x_sfc_n || !y_lev_n || !y_sfc_n || !x_lev_denorm ||
        !y_lev_denorm || !y_sfc_denorm) {
        csa_set_error_msg("csa_gen_batch: bad argument");
        return CSA_ERR_ARG;
    }
    const long tot = (long)N * h->d.c.nlev;
    const size_t shm = sizeof(float) * GEN_T * ((h->d.nx_out <= 16 ? 17 : 33) + GEN_SY + 1);
    hipLaunchKernelGGL(gen_batch_kernel, dim3((unsigned)((tot + GEN_T - 1) / GEN_T)), dim3(GEN_T), shm, (hipStream_t)stream, h->d, N, x_lev, x_sfc, y_lev,
                       y_sfc, x_lev_n, x_sfc_n, y_lev_n, y_sfc_n, x_lev_denorm, y_lev_denorm, y_sfc_denorm);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
