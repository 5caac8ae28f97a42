// phys.hip -- the reference's physRNN "Hidden" model, forward, as shipped in
// rnn/saved_models/physRNN-Hidden_*_script_cpu.pt:
//   physical_RNN_autoreg.forward            rnn/models/models_phys.py:1586-1823
//   physical_RNN_autoreg.microphysics_decode rnn/models/models_phys.py:414-748
//   LayerPressure / thickness / level pressure rnn/layers.py:117-168
// BiGRU core (nx = 21 level inputs + sqrt(p)/314, GRU up over [tanh(mlp_initial) | 15 memory channels, zero above the
// CRM top ilev_crm], GRU down from a caller-supplied N(0,1) state), then a physically constrained decoder: eleven
// Linear(nh, mp_ncol) heads describe mp_ncol sub-grid columns per level (vapour, condensate, ice, temperature offsets,
// area fractions, mass flux, eddy diffusivity, sedimentation, evaporation, condensation, autoconversion); tendencies
// are flux divergences on the pressure grid with positivity clamps, area-weighted back to the grid column; precipitation
// is a column water budget with a stored-water memory channel.
//
// Launches per call (one stream): phys_prep_kernel -> projection GEMM -> GRU recurrence (rec.hip) -> projection GEMM ->
// GRU recurrence -> ONE head GEMM (192 = 11 x 16 decoder heads + 15 latent + 1 radiative heating columns, K = nh) ->
// phys_decode_kernel (one workgroup per grid column: all vertical differences, level softmax and column sums in LDS).
// Activations are level-major (level, column, channel) as in the rest of the library.
#include "common.h"
#include "pack.h"
#include <vector>

#define PH_L 60
#define PH_NCOL 16
#define PH_NHEAD 11
#define PH_HD 192          // 11*16 + 15 + 1

struct PhysDev {
    int nx, nx_sfc, nh, ilev, nm0, Lc;
    const float *hyam, *hybm, *hyai, *hybi, *yscale_lev, *yscale_sca;
    float xdiv_sca0, xmean_sca0;
    const float *init_wt, *init_b, *s1_wt, *s1_b;   // (nx+1, nh), (nx_sfc, nh) transposed
    const float *out_w, *out_b;                     // mlp_output (5, nm0)
    const float *sfo_w, *sfo_b;                     // mlp_surface_output_rad (6, nh)
    const float *rel_w, *rel_b;                     // mlp_precip_release (1, nh)
};

struct csa_phys {
    PhysDev d;
    int max_batch;
    float *wih1, *bias1, *bhn1, *whh1p, *whh1g, *wih2, *bias2, *bhn2, *whh2p, *whh2g, *whead, *bhead;
    float *X1, *P, *H1, *H2, *hx, *HD;
    std::vector<void *> owned;
};

__device__ __forceinline__ float ph_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }   // torch.softplus(beta 1, threshold 20)
__device__ __forceinline__ float ph_sum16(float v)
{
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ float ph_max16(float v)
{
    v = fmaxf(v, __shfl_xor(v, 8)); v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 1));
    return v;
}

// one workgroup (128 threads = nh) per grid column: thread j owns hidden unit j of mlp_initial / mlp_surface1
__global__ __launch_bounds__(128) void phys_prep_kernel(PhysDev d, int B, const float *__restrict__ x_main, const float *__restrict__ x_sfc,
                                                        const float *__restrict__ mem, float *__restrict__ X1, float *__restrict__ hx)
{
    // [L][32] inputs incl. the pressure feature, zero-padded to 32 per level; then [64] surface inputs, zero-padded.
    // The padding makes every inner loop below branch-free with a compile-time trip count: with the runtime bound
    // (nx + 1 = 22) the compiler emitted one scalar branch and one s_waitcnt per LDS read -- 80 us instead of 13.
    __shared__ float xin[PH_L * 32];
    __shared__ float xs[64];
    const int b = blockIdx.x, j = threadIdx.x, nx1 = d.nx + 1, nh = d.nh, K1 = nh + 16;
    // the levels are independent here: blockIdx.y takes one half of them (384 columns alone are 1.5 workgroups per CU)
    const int lper = (PH_L + (int)gridDim.y - 1) / (int)gridDim.y, l0 = blockIdx.y * lper, l1 = min(PH_L, l0 + lper);
    for (int i = j; i < PH_L * 32; i += 128) xin[i] = 0.0f;
    if (j < 64) xs[j] = j < d.nx_sfc ? x_sfc[(size_t)b * d.nx_sfc + j] : 0.0f;
    __syncthreads();
    for (int i = l0 * d.nx + j; i < l1 * d.nx; i += 128) {
        const int l = i / d.nx, v = i - l * d.nx;
        xin[l * 32 + v] = x_main[(size_t)b * PH_L * d.nx + i];
    }
    const float sp = xs[0] * d.xdiv_sca0 + d.xmean_sca0;
    for (int l = l0 + j; l < l1; l += 128) xin[l * 32 + d.nx] = sqrtf(d.hyam[l] * 100000.0f + sp * d.hybm[l]) / 314.0f;
    __syncthreads();
    if (j < nh) {
        if (blockIdx.y == 0) {
            float a = d.s1_b[j];
            for (int k0 = 0; k0 < d.nx_sfc; k0 += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = d.s1_wt[min(k0 + u, d.nx_sfc - 1) * nh + j];     // clamped index: xs is zero there
#pragma unroll
                for (int u = 0; u < 8; ++u) a = fmaf(xs[k0 + u], t[u], a);
            }
            hx[(size_t)b * nh + j] = tanhf(a);
        }
        float w[32];
        const float bj = d.init_b[j];
#pragma unroll
        for (int k = 0; k < 32; ++k) w[k] = d.init_wt[min(k, nx1 - 1) * nh + j];                  // xin is zero for k >= nx + 1
        for (int l = l0; l < l1; ++l) {
            float acc = bj;
#pragma unroll
            for (int k = 0; k < 32; ++k) acc = fmaf(xin[l * 32 + k], w[k], acc);
            X1[((size_t)(PH_L - 1 - l) * B + b) * K1 + j] = tanhf(acc);      // rnn1 runs over the flipped level axis
        }
    }
    // memory channels (15 carried + zero pad to 16), zero above the CRM top
    for (int i = l0 * 16 + j; i < l1 * 16; i += 128) {
        const int l = i >> 4, k = i & 15;
        const float v = (l >= d.ilev && k < d.nm0) ? mem[((size_t)b * d.Lc + (l - d.ilev)) * (d.nm0 + 1) + k] : 0.0f;
        X1[((size_t)(PH_L - 1 - l) * B + b) * K1 + nh + k] = v;
    }
}

// head-GEMM column order
enum { H_QV = 0, H_QN, H_T, H_AREA, H_FLUX, H_EDDY, H_QICE, H_SED, H_EVAP, H_COND, H_AA };

#define PH_DT 512          // decoder workgroup: 800 (level, sub-column) cells in two passes
__global__ __launch_bounds__(PH_DT) void phys_decode_kernel(PhysDev d, int B, const float *__restrict__ HD, const float *__restrict__ H2,
                                                          const float *__restrict__ x_sfc, const float *__restrict__ mem,
                                                          const float *__restrict__ x_denorm, int nxd,
                                                          float *__restrict__ out_lev, float *__restrict__ out_sfc, float *__restrict__ mem_out)
{
    constexpr int LC = 50, NC = PH_NCOL;
    __shared__ float s_out[LC][5], s_pv[LC], s_pd[LC], s_dprec[LC], s_red[8];
    __shared__ float s_area[LC * NC], s_qv[LC * NC], s_qn[LC * NC], s_fH[LC * NC], s_fqv[LC * NC], s_fqn[LC * NC], s_sed[LC * NC];
    __shared__ float s_scal[16];
    constexpr int nh = 128, nm0 = 15;                 // enforced by csa_phys_create: compile-time trip counts (see phys_prep_kernel)
    const int b = blockIdx.x, tid = threadIdx.x, ilev = d.ilev;
    const float CP = 1004.64f, G = 9.80665f, LV = 2510400.0f, LS = 2844000.0f, OOG = 0.1019716213f;
    const float sp = x_sfc[(size_t)b * d.nx_sfc] * d.xdiv_sca0 + d.xmean_sca0;
    const float P_old = mem[((size_t)b * LC + (LC - 1)) * (nm0 + 1) + nm0];
    const float *last_h = H2 + ((size_t)(PH_L - 1) * B + b) * nh;

    // ---- phase A: latent memory -> mlp_output per level; level pressure thickness; surface heads ----
    for (int l = tid; l < LC; l += PH_DT) {
        const float *hd = HD + ((size_t)(l + ilev) * B + b) * PH_HD + PH_NHEAD * NC;
        float lat[16];
#pragma unroll
        for (int k = 0; k < nm0; ++k) { lat[k] = hd[k]; mem_out[((size_t)b * LC + l) * (nm0 + 1) + k] = lat[k]; }
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            float a = d.out_b[v];
#pragma unroll
            for (int k = 0; k < nm0; ++k) a = fmaf(lat[k], d.out_w[v * nm0 + k], a);
            s_out[l][v] = a;
        }
        const int L = l + ilev;
        s_pd[l] = sp * (d.hybi[L + 1] - d.hybi[L]) + (d.hyai[L + 1] - d.hyai[L]) * 100000.0f;
    }
    if (tid >= 64 && tid < 64 + 7 * 8) {              // 6 radiative surface outputs + the precipitation release logit: 8 lanes each
        const int o = (tid - 64) >> 3, part = tid & 7;
        const float *w = o < 6 ? d.sfo_w + o * nh : d.rel_w;
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < nh / 8; ++k) a = fmaf(last_h[part + 8 * k], w[part + 8 * k], a);
        a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
        if (part == 0) s_scal[o] = a + (o < 6 ? d.sfo_b[o] : d.rel_b[0]);
    }
    __syncthreads();
    // softmax over the 50 levels of out[:, :, 2], times the stored water  (first wave)
    if (tid < 64) {
        const float v = tid < LC ? s_out[tid][2] : -3.0e38f;
        float m = v;
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float e = tid < LC ? expf(v - m) : 0.0f;
        float s = e;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (tid < LC) s_pv[tid] = e / s * P_old;
    }

    // ---- phase B: sub-grid state and the fluxes at each level ----
    for (int e0 = 0; e0 < LC * NC; e0 += PH_DT) {
        const int e = e0 + tid, l = e >> 4, c = e & 15;
        const bool ok = l < LC;
        const int L = (ok ? l : 0) + ilev;
        const float *hd = HD + ((size_t)L * B + b) * PH_HD + c;
        const float *xd = x_denorm + ((size_t)b * PH_L + L) * nxd;
        const float a_raw = hd[H_AREA * NC];
        const float am = ph_max16(a_raw), ae = expf(a_raw - am), area = ae / ph_sum16(ae);
        float qv = ph_softplus(hd[H_QV * NC]), qn = ph_softplus(hd[H_QN * NC]), qi = ph_softplus(hd[H_QICE * NC]);
        const float mqv = ph_sum16(qv * area), mqn = ph_sum16(qn * area), mqi = ph_sum16(qi * area);
        qv *= mqv == 0.0f ? 1.0f : xd[nxd - 1] / mqv;
        qn *= mqn == 0.0f ? 1.0f : (xd[2] + xd[3]) / mqn;
        qi *= mqi == 0.0f ? 1.0f : xd[3] / mqi;
        const float dT = hd[H_T * NC];
        const float T_crm = xd[0] + (dT - ph_sum16(dT * area));
        const float play = d.hyam[L] * 100000.0f + sp * d.hybm[L], play_up = d.hyam[L - 1] * 100000.0f + sp * d.hybm[L - 1];
        float fH = hd[H_EDDY * NC] * (CP / G) * T_crm * (play - play_up);
        if (l == LC - 1) fH = -fmaxf(fH, 0.0f);
        const float flux1 = hd[H_FLUX * NC] * 300000.0f;
        if (ok) {
            s_area[e] = area; s_qv[e] = qv; s_qn[e] = qn; s_fH[e] = fH;
            s_fqv[e] = flux1 * qv; s_fqn[e] = flux1 * qn;
            s_sed[e] = fmaxf(hd[H_SED * NC], 0.0f) * G * qi * d.yscale_lev[L * 5 + 2];
        }
    }
    __syncthreads();

    // ---- phase C: flux divergences, clamps, tendencies, area-weighted means ----
    for (int e0 = 0; e0 < LC * NC; e0 += PH_DT) {
        const int e = e0 + tid, l = e >> 4, c = e & 15;
        const bool ok = l < LC;
        const int lc = ok ? l : 0, ec = ok ? e : c, L = lc + ilev;
        const float *hd = HD + ((size_t)L * B + b) * PH_HD + c;
        const float *ys = d.yscale_lev + L * 5;
        const float pd = s_pd[lc], area = s_area[ec], qv = s_qv[ec], qn = s_qn[ec];
        const bool up = lc > 0, last = lc == LC - 1;
        const float flux_t_dp = (s_fH[ec] - (up ? s_fH[ec - NC] : 0.0f)) / pd * (-G / CP);
        const float flux_qv_dp = ((last ? 0.0f : s_fqv[ec]) - (up ? s_fqv[ec - NC] : 0.0f)) / pd * (-G);
        const float flux_qn_dp = ((last ? 0.0f : s_fqn[ec]) - (up ? s_fqn[ec - NC] : 0.0f)) / pd * (-G);
        const float sed_qn_dp = (s_sed[ec] - (up ? s_sed[ec - NC] : 0.0f)) / pd * (-G);
        float evap = (fmaxf(hd[H_EVAP * NC], 0.0f) + 1e-6f) * s_pv[lc];
        float cond = hd[H_COND * NC];
        float aa = fmaxf(hd[H_AA * NC], 0.0f) * qn * ys[2];
        cond = fmaxf(cond, ((-(ys[2] * qn / 1200.0f) - flux_qn_dp) + aa) - sed_qn_dp);
        evap = fmaxf(evap, (-(ys[1] * qv / 1200.0f) - flux_qv_dp) + cond);
        aa = fmaxf(aa, ((flux_qn_dp + cond) + sed_qn_dp) - ys[2] * (-qn + 0.0006f) / 1200.0f);
        const float dqv = (flux_qv_dp - cond) + evap;
        const float dqn = ((flux_qn_dp + cond) - aa) + sed_qn_dp;
        const float *xd = x_denorm + ((size_t)b * PH_L + L) * nxd;
        const float temp = xd[0] + (ph_sum16(area * flux_t_dp) / ys[0]) * 1200.0f;
        const float liq = fminf(fmaxf((temp - 253.16f) * 0.05f, 0.0f), 1.0f);
        const float net = ((liq * LV + (1.0f - liq) * LS) * cond - evap * LV) * (1.0f / CP);
        const float dT_crm = flux_t_dp + net / ys[1] * ys[0];
        const float sT = ph_sum16(area * dT_crm), sqv = ph_sum16(area * dqv), sqn = ph_sum16(area * dqn);
        const float sprec = ph_sum16(area * (aa - evap));
        const float ssed = ph_sum16(area * s_sed[ec]);
        if (ok && c == 0) {
            const float dT_rad = HD[((size_t)L * B + b) * PH_HD + PH_HD - 1];
            float *o = out_lev + ((size_t)b * PH_L + L) * 5;
            o[0] = ((l >= 2 ? s_out[l][0] : 0.0f) + sT) + dT_rad;
            o[1] = sqv;
            o[2] = sqn;
            o[3] = l >= 2 ? s_out[l][3] : 0.0f;
            o[4] = l >= 2 ? s_out[l][4] : 0.0f;
            s_dprec[l] = pd * OOG * sprec;
            if (last) s_scal[8] = ssed;               // sedimentation reaching the surface
        }
    }
    // levels above the CRM top: only the radiative heating
    for (int L = tid; L < ilev; L += PH_DT) {
        float *o = out_lev + ((size_t)b * PH_L + L) * 5;
        o[0] = HD[((size_t)L * B + b) * PH_HD + PH_HD - 1];
        o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f; o[4] = 0.0f;
    }
    __syncthreads();

    // ---- phase D: column water budget, precipitation, stored-water memory channel ----
    if (tid < 64) {
        float w = tid < LC ? s_dprec[tid] : 0.0f;
        for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
        if (tid == 0) {
            const float water_new = fmaxf(P_old + w, 0.0f);
            const float rel = 1.0f / (1.0f + expf(-s_scal[6]));
            const float released = rel * water_new;
            float stored = water_new * (1.0f - rel);
            const float Tsfc = x_denorm[((size_t)b * PH_L + (PH_L - 1)) * nxd];
            const float Pmax = d.yscale_sca[3] * 1000.0f * 5.58e-18f * expf(Tsfc * 0.077f);
            const float excess = fmaxf(stored - Pmax, 0.0f);
            stored -= excess;
            const float precc = ((s_scal[8] + released) + excess) / 1000.0f;
            const float snowfrac = fminf(fmaxf((-Tsfc + 283.3f) / 14.6f, 0.0f), 1.0f);
            float *os = out_sfc + (size_t)b * 8;
            os[0] = fmaxf(s_scal[0], 0.0f); os[1] = fmaxf(s_scal[1], 0.0f);
            os[2] = snowfrac * precc; os[3] = precc;
            for (int k = 2; k < 6; ++k) os[2 + k] = fmaxf(s_scal[k], 0.0f);
            s_red[0] = stored;
        }
    }
    __syncthreads();
    for (int l = tid; l < LC; l += PH_DT) mem_out[((size_t)b * LC + l) * (nm0 + 1) + nm0] = s_red[0];
}

// ------------------------------------------------------------------------------------------------------------------
extern "C" int csa_phys_create(int nx, int nx_sfc, int nh, int ilev_crm, int mp_ncol, int nh_mem0,
                               const float *const *w /* see include/climsim_amd.h for the order */, int max_batch, csa_phys **out)
{
    if (!w || !out || max_batch <= 0) { csa_set_error_msg("csa_phys_create: bad argument"); return CSA_ERR_ARG; }
    if (nh != 128 || mp_ncol != PH_NCOL || nh_mem0 != 15 || ilev_crm != 10 || nx + 1 > 32 || nx_sfc > 64) {
        csa_set_error_msg("csa_phys_create: built for the shipped physRNN-Hidden geometry (GRU 128/128, mp_ncol 16, 15+1 memory channels, ilev_crm 10)");
        return CSA_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { csa_set_error_msg("csa_phys_create: no HIP device"); return CSA_ERR_HIP; }
    csa_phys *h = new csa_phys();
    h->max_batch = max_batch;
    int rc = CSA_OK;
    auto up = [&](const float *src, size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, sizeof(float) * n) != hipSuccess) { rc = CSA_ERR_NOMEM; return (float *)nullptr; }
        h->owned.push_back(p);
        if (src && hipMemcpy(p, src, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
        return (float *)p;
    };
    enum { W_HYAM, W_HYBM, W_HYAI, W_HYBI, W_YSL, W_YSS, W_XDS, W_XMS, W_INIT_W, W_INIT_B, W_S1_W, W_S1_B, W_R1_IH, W_R1_HH, W_R1_BIH,
           W_R1_BHH, W_R2_IH, W_R2_HH, W_R2_BIH, W_R2_BHH, W_LAT_W, W_LAT_B, W_OUT_W, W_OUT_B, W_SFO_W, W_SFO_B, W_RAD_W, W_RAD_B,
           W_REL_W, W_REL_B, W_HEADS /* 11 x (weight, bias) in head-GEMM column order */ };
    PhysDev &d = h->d;
    d.nx = nx; d.nx_sfc = nx_sfc; d.nh = nh; d.ilev = ilev_crm; d.nm0 = nh_mem0; d.Lc = PH_L - ilev_crm;
    d.hyam = up(w[W_HYAM], 60); d.hybm = up(w[W_HYBM], 60); d.hyai = up(w[W_HYAI], 61); d.hybi = up(w[W_HYBI], 61);
    d.yscale_lev = up(w[W_YSL], 60 * 5); d.yscale_sca = up(w[W_YSS], 8);
    d.xdiv_sca0 = w[W_XDS][0]; d.xmean_sca0 = w[W_XMS][0];
    { auto t = transposed(w[W_INIT_W], nh, nx + 1); d.init_wt = up(t.data(), t.size()); }
    d.init_b = up(w[W_INIT_B], nh);
    { auto t = transposed(w[W_S1_W], nh, nx_sfc); d.s1_wt = up(t.data(), t.size()); }
    d.s1_b = up(w[W_S1_B], nh);
    d.out_w = up(w[W_OUT_W], 5 * nh_mem0); d.out_b = up(w[W_OUT_B], 5);
    d.sfo_w = up(w[W_SFO_W], 6 * nh); d.sfo_b = up(w[W_SFO_B], 6);
    d.rel_w = up(w[W_REL_W], nh); d.rel_b = up(w[W_REL_B], 1);
    // GRU layers: W_ih rows to unit-major [r, z, n, 0]; rnn1's K = nh + 15 padded to nh + 16 with a zero column
    {
        const int Kin = nh + nh_mem0, K1 = nh + 16;
        std::vector<float> wpad((size_t)3 * nh * K1, 0.0f), wp, bp, bhn;
        for (int r = 0; r < 3 * nh; ++r) memcpy(&wpad[(size_t)r * K1], &w[W_R1_IH][(size_t)r * Kin], sizeof(float) * Kin);
        pack_ih(0, nh, K1, wpad.data(), w[W_R1_BIH], w[W_R1_BHH], wp, bp, bhn);
        h->wih1 = up(wp.data(), wp.size()); h->bias1 = up(bp.data(), bp.size()); h->bhn1 = up(bhn.data(), bhn.size());
        pack_ih(0, nh, nh, w[W_R2_IH], w[W_R2_BIH], w[W_R2_BHH], wp, bp, bhn);
        h->wih2 = up(wp.data(), wp.size()); h->bias2 = up(bp.data(), bp.size()); h->bhn2 = up(bhn.data(), bhn.size());
        std::vector<float> pk(rec_packed_floats(0, nh));
        rec_pack_weights(0, nh, w[W_R1_HH], pk.data()); h->whh1p = up(pk.data(), pk.size());
        rec_pack_weights(0, nh, w[W_R2_HH], pk.data()); h->whh2p = up(pk.data(), pk.size());
        gru2_pack_weights(nh, w[W_R1_HH], pk.data()); h->whh1g = up(pk.data(), pk.size());   // two-column kernel (B > 256)
        gru2_pack_weights(nh, w[W_R2_HH], pk.data()); h->whh2g = up(pk.data(), pk.size());
    }
    // head GEMM: 11 decoder heads (16 rows each), mlp_latent (15 rows), mlp_output_rad (1 row)
    {
        std::vector<float> wh((size_t)PH_HD * nh), bh(PH_HD);
        for (int k = 0; k < PH_NHEAD; ++k) {
            memcpy(&wh[(size_t)k * PH_NCOL * nh], w[W_HEADS + 2 * k], sizeof(float) * PH_NCOL * nh);
            memcpy(&bh[k * PH_NCOL], w[W_HEADS + 2 * k + 1], sizeof(float) * PH_NCOL);
        }
        memcpy(&wh[(size_t)PH_NHEAD * PH_NCOL * nh], w[W_LAT_W], sizeof(float) * nh_mem0 * nh);
        memcpy(&bh[PH_NHEAD * PH_NCOL], w[W_LAT_B], sizeof(float) * nh_mem0);
        memcpy(&wh[(size_t)(PH_HD - 1) * nh], w[W_RAD_W], sizeof(float) * nh);
        bh[PH_HD - 1] = w[W_RAD_B][0];
        h->whead = up(wh.data(), wh.size()); h->bhead = up(bh.data(), bh.size());
    }
    const size_t rows = (size_t)PH_L * max_batch;
    h->X1 = up(nullptr, rows * (nh + 16)); h->P = up(nullptr, rows * 4 * nh); h->H1 = up(nullptr, rows * nh);
    h->H2 = up(nullptr, rows * nh); h->hx = up(nullptr, (size_t)max_batch * nh); h->HD = up(nullptr, rows * PH_HD);
    if (rc != CSA_OK) { for (void *p : h->owned) (void)hipFree(p); delete h; return rc; }
    *out = h;
    return CSA_OK;
}

extern "C" int csa_phys_destroy(csa_phys *h)
{
    if (!h) return CSA_ERR_ARG;
    for (void *p : h->owned) (void)hipFree(p);
    delete h;
    return CSA_OK;
}

// x_main (B,60,nx) normalised, x_sfc (B,nx_sfc) normalised, rnn_mem (B,50,16), x_denorm (B,60,nxd) raw (T, ., qliq, qice, ..., qv last),
// hx2 (B,nh): the N(0,1) draw the reference makes for rnn2's initial state.  -> out_lev (B,60,5), out_sfc (B,8), mem_out (B,50,16)
extern "C" int csa_phys_forward(csa_phys *h, int B, const float *x_main, const float *x_sfc, const float *rnn_mem,
                                const float *x_denorm, int nxd, const float *hx2, float *out_lev, float *out_sfc, float *mem_out,
                                void *stream)
{
    if (!h || !x_main || !x_sfc || !rnn_mem || !x_denorm || !hx2 || !out_lev || !out_sfc || !mem_out || B <= 0 || B > h->max_batch || nxd < 5) {
        csa_set_error_msg("csa_phys_forward: bad argument");
        return CSA_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const PhysDev &d = h->d;
    const int nh = d.nh, M = PH_L * B;
    int rc;
    hipLaunchKernelGGL(phys_prep_kernel, dim3(B, B <= 1024 ? 2 : 1), dim3(128), 0, s, d, B, x_main, x_sfc, rnn_mem, h->X1, h->hx);
    CSA_HIP_CHECK(hipGetLastError());
    auto rec = [&](const float *whh, const float *whg, const float *bhn, const float *h0, float *Hout, int reverse) {
        return B <= 256 ? launch_rec1_gru(nh, whh, bhn, h->P, h0, Hout, B, PH_L, reverse, s)
                        : launch_rec2_gru(nh, whg, bhn, h->P, h0, Hout, B, PH_L, reverse, s);
    };
    if ((rc = launch_proj_gemm(h->X1, h->wih1, h->bias1, h->P, M, 4 * nh, nh + 16, s, 0))) return rc;
    if ((rc = rec(h->whh1p, h->whh1g, h->bhn1, h->hx, h->H1, 1))) return rc;
    if ((rc = launch_proj_gemm(h->H1, h->wih2, h->bias2, h->P, M, 4 * nh, nh, s, 0))) return rc;
    if ((rc = rec(h->whh2p, h->whh2g, h->bhn2, hx2, h->H2, 0))) return rc;
    if ((rc = launch_proj_gemm(h->H2, h->whead, h->bhead, h->HD, M, PH_HD, nh, s, 0))) return rc;
    hipLaunchKernelGGL(phys_decode_kernel, dim3(B), dim3(PH_DT), 0, s, d, B, h->HD, h->H2, x_sfc, rnn_mem, x_denorm, nxd, out_lev, out_sfc, mem_out);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// taps for tests: level-major (60, B, nh) outputs of rnn1 (level order) and rnn2 of the last call
extern "C" int csa_phys_tap(csa_phys *h, int which, int B, float *dst, void *stream)
{
    if (!h || !dst || B <= 0 || B > h->max_batch || which < 1 || which > 2) return CSA_ERR_ARG;
    CSA_HIP_CHECK(hipMemcpyAsync(dst, which == 1 ? h->H1 : h->H2, sizeof(float) * (size_t)PH_L * B * h->d.nh, hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
    return CSA_OK;
}
