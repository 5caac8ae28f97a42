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
//
// Two serialised geometries (csa_phys_create / csa_phys_rad_create):
//   "mp16": 21 level inputs (+ pressure) on all 60 levels, 19 surface inputs, mp_ncol 16, radiative heating from two
//           Linear heads (num14564, num49672)
//   "rad" : 21 level inputs of which the first 18 (+ pressure) feed mlp_initial, both GRUs over the 50 CRM levels only,
//           surface inputs 0:6 and 11:19, mp_ncol 4, and the physical radiation scheme of phys_rad.hip after the
//           decoder (num4050)
#include "phys.h"
#include "stoch.h"
// one workgroup (128 threads = nh) per grid column: thread j owns hidden unit j of mlp_initial / mlp_surface1
__global__ __launch_bounds__(128) void phys_prep_kernel(PhysDev d, int B, const float *__restrict__ x_main, const float *__restrict__ x_sfc,
                                                        const float *__restrict__ mem, float *__restrict__ X1, float *__restrict__ hx)
{
    // [L][32] inputs incl. the pressure feature, zero-padded to 32 per level; then [64] surface inputs, zero-padded.
    // The padding makes every inner loop below branch-free with a compile-time trip count: with the runtime bound
    // (nx + 1 = 22) the compiler emitted one scalar branch and one s_waitcnt per LDS read -- 80 us instead of 13.
    __shared__ float xin[PH_L * 32];
    __shared__ float xs[64];
    const int b = blockIdx.x, j = threadIdx.x, nf1 = d.nfeat + 1, nh = d.nh, K1 = nh + 16;
    // the levels are independent here: blockIdx.y takes one half of them (384 columns alone are 1.5 workgroups per CU)
    const int lper = (d.Lr + (int)gridDim.y - 1) / (int)gridDim.y, l0 = d.ltop + blockIdx.y * lper, l1 = min(PH_L, l0 + lper);
    for (int i = j; i < PH_L * 32; i += 128) xin[i] = 0.0f;
    if (j < 64) xs[j] = j < d.nx_sfc ? x_sfc[(size_t)b * d.naux + (j < d.sfc_cut ? j : j + d.sfc_skip)] : 0.0f;
    const float sp = x_sfc[(size_t)b * d.naux] * d.xdiv_sca0 + d.xmean_sca0;
    __syncthreads();
    for (int i = l0 * d.nx + j; i < l1 * d.nx; i += 128) {
        const int l = i / d.nx, v = i - l * d.nx;
        if (v < d.nfeat) xin[l * 32 + v] = x_main[(size_t)b * PH_L * d.nx + i];
    }
    for (int l = l0 + j; l < l1; l += 128) xin[l * 32 + d.nfeat] = sqrtf(d.hyam[l] * 100000.0f + sp * d.hybm[l]) / 314.0f;
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
        for (int k = 0; k < 32; ++k) w[k] = d.init_wt[min(k, nf1 - 1) * nh + j];                  // xin is zero for k >= nfeat + 1
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
        const float v = (l >= d.ilev && k < d.nm0) ? mem[ph_mem_row(d, B, b, l - d.ilev) * (d.nm0 + 1) + k] : 0.0f;
        X1[((size_t)(PH_L - 1 - l) * B + b) * K1 + nh + k] = v;
    }
}


// E3SM's ice effective radius table and liquid effective radius rule (rnn/models/physics_rad_e3sm.py:13, :62)
__device__ __forceinline__ float ph_reitab(const float *__restrict__ tab, float t)
{
    const int i = min(max((int)(t - 136.0f), 1), PH_NRETAB - 2);
    const float w = t - floorf(t);
    return tab[i] * (1.0f - w) + tab[i + 1] * w;
}
__device__ __forceinline__ float ph_reltab(float t, float landfrac, float icefrac, float snowh)
{
    float rel = 8.0f + 6.0f * fminf(fmaxf((273.15f - t) * 0.05f, 0.0f), 1.0f);
    rel = rel + (14.0f - rel) * fminf(fmaxf(snowh * 10.0f, 0.0f), 1.0f);
    rel = rel + (14.0f - rel) * fminf(fmaxf(1.0f - landfrac, 0.0f), 1.0f);
    return rel + (14.0f - rel) * fminf(fmaxf(icefrac, 0.0f), 1.0f);
}

// Radiation work arrays written by the decoder when RAD (consumed by phys_rad.hip); rows are (level, column), level-major
struct PhysRadOut {
    const float *x_main;
    float *XG, *XR, *RS, *CL, *CS;
};

// decoder workgroup: DT threads over the LC * NC (level, sub-column) cells -- 800 cells in two passes of 512 for
// mp_ncol 16, 200 cells in one pass of 256 for mp_ncol 4
template <int NC, int DT, bool RAD>
__global__ __launch_bounds__(DT) void phys_decode_kernel(PhysDev d, int B, const float *__restrict__ HD, const float *__restrict__ H2,
                                                         const float *__restrict__ x_sfc, const float *__restrict__ mem,
                                                         const float *__restrict__ x_denorm, int nxd,
                                                         float *__restrict__ out_lev, float *__restrict__ out_sfc, float *__restrict__ mem_out,
                                                         PhysRadOut ro)
{
    constexpr int LC = 50, SH = NC == 16 ? 4 : 2;
    static_assert(NC == 16 || NC == 4, "mp_ncol");
    __shared__ float s_out[LC][5], s_pv[LC], s_pd[LC], s_dprec[LC], s_red[8];
    __shared__ float s_area[LC * NC], s_qv[LC * NC], s_qn[LC * NC], s_fH[LC * NC], s_fqv[LC * NC], s_fqn[LC * NC], s_sed[LC * NC];
    __shared__ float s_T[LC * NC];                    // sub-column temperature (RAD: overwritten with its updated value)
    __shared__ float s_liq[RAD ? LC * NC : 1];        // nx21: the region's cloud liquid fraction (latent heating AND cloud optics)
    __shared__ float s_o01[PH_L][3];                  // RAD: the decoder's dT, dqv, dqn of every level (zero above the CRM top)
    __shared__ float s_scal[16];
    constexpr int nm0 = 15;                           // enforced by csa_phys_create: compile-time trip counts (see phys_prep_kernel)
    const int nh = d.nh;
    const int b = blockIdx.x, tid = threadIdx.x, ilev = d.ilev, hd0 = ilev - d.ltop, HDW = d.hdw;
    const float CP = 1004.64f, G = 9.80665f, LV = 2510400.0f, LS = 2844000.0f, OOG = 0.1019716213f;
    const float sp = x_sfc[(size_t)b * d.naux] * d.xdiv_sca0 + d.xmean_sca0;
    const float P_old = mem[ph_mem_row(d, B, b, LC - 1) * (nm0 + 1) + nm0];
    const float *last_h = H2 + ((size_t)(d.Lr - 1) * B + b) * nh;      // (H2: whichever sequence ends in the state the release head reads)

    // ---- phase A: latent memory -> mlp_output per level; level pressure thickness; surface heads ----
    for (int l = tid; l < LC; l += DT) {
        const float *hd = HD + ((size_t)(l + hd0) * B + b) * HDW + PH_NHEAD * NC;
        float lat[16];
#pragma unroll
        for (int k = 0; k < nm0; ++k) { lat[k] = hd[k]; mem_out[ph_mem_row(d, B, b, l) * (nm0 + 1) + k] = lat[k]; }
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
        if (!RAD || o == 6) {
            const float *w = o < 6 ? d.sfo_w + o * nh : d.rel_w;
            float a = 0.0f;
            for (int k = 0; k < nh / 8; ++k) a = fmaf(last_h[part + 8 * k], w[part + 8 * k], a);
            a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
            if (part == 0) s_scal[o] = a + (o < 6 ? d.sfo_b[o] : d.rel_b[0]);
        }
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
    // (the head-GEMM and input values of ALL passes are requested before the first is used: the shuffles below are convergent
    //  operations the compiler does not move loads across, and a pass otherwise starts with a full memory round trip)
    constexpr int NPASS = (LC * NC + DT - 1) / DT;
    float hB[NPASS][8], xB[NPASS][4], hC[NPASS][3];      // (hC: the three heads phase C reads, same rows)
#pragma unroll
    for (int ip = 0; ip < NPASS; ++ip) {
        const int e = ip * DT + tid, l = e >> SH, c = e & (NC - 1);
        const int L = (l < LC ? l : 0) + ilev;
        const float *hd = HD + ((size_t)(L - d.ltop) * B + b) * HDW + c;
        const float *xd = x_denorm + ((size_t)b * PH_L + L) * nxd;
        hB[ip][0] = hd[H_AREA * NC]; hB[ip][1] = hd[H_QV * NC]; hB[ip][2] = hd[H_QN * NC]; hB[ip][3] = hd[H_QICE * NC];
        hB[ip][4] = hd[H_T * NC]; hB[ip][5] = hd[H_EDDY * NC]; hB[ip][6] = hd[H_FLUX * NC]; hB[ip][7] = hd[H_SED * NC];
        xB[ip][0] = xd[0]; xB[ip][1] = xd[2]; xB[ip][2] = xd[3]; xB[ip][3] = xd[nxd - 1];
        hC[ip][0] = hd[H_EVAP * NC]; hC[ip][1] = hd[H_COND * NC]; hC[ip][2] = hd[H_AA * NC];
    }
#pragma unroll
    for (int ip = 0; ip < NPASS; ++ip) {
        const int e = ip * DT + tid, l = e >> SH, c = e & (NC - 1);
        const bool ok = l < LC;
        const int L = (ok ? l : 0) + ilev;
        const float a_raw = hB[ip][0];
        const float am = ph_max<NC>(a_raw), ae = expf(a_raw - am), area = ae / ph_sum<NC>(ae);
        float qv = ph_softplus(hB[ip][1]), qn = ph_softplus(hB[ip][2]), qi = ph_softplus(hB[ip][3]);
        if (d.clear0 && c == 0) qn = 0.0f;           // clear-sky region (its condensation head row is zero too: see phys_build)
        const float mqv = ph_sum<NC>(qv * area), mqn = ph_sum<NC>(qn * area), mqi = ph_sum<NC>(qi * area);
        qv *= mqv == 0.0f ? 1.0f : xB[ip][3] / mqv;
        qn *= mqn == 0.0f ? 1.0f : (xB[ip][1] + xB[ip][2]) / mqn;
        qi *= mqi == 0.0f ? 1.0f : xB[ip][2] / mqi;
        const float dT = hB[ip][4];
        const float T_crm = xB[ip][0] + (dT - ph_sum<NC>(dT * area));
        const float play = d.hyam[L] * 100000.0f + sp * d.hybm[L], play_up = d.hyam[L - 1] * 100000.0f + sp * d.hybm[L - 1];
        float fH = hB[ip][5] * (CP / G) * T_crm * (play - play_up);
        if (l == LC - 1 && !d.dec21) fH = -fmaxf(fH, 0.0f);     // (nx21: defined at layer tops like the moisture fluxes, zero at the surface)
        const float flux1 = hB[ip][6] * 300000.0f;
        if (ok) {
            s_area[e] = area; s_qv[e] = qv; s_qn[e] = qn; s_fH[e] = fH; s_T[e] = T_crm;
            s_fqv[e] = flux1 * qv; s_fqn[e] = flux1 * qn;
            s_sed[e] = fmaxf(hB[ip][7], 0.0f) * G * qi * d.yscale_lev[L * 5 + 2];
        }
    }
    __syncthreads();

    // ---- phase C: flux divergences, clamps, tendencies, area-weighted means ----
#pragma unroll
    for (int ip = 0; ip < NPASS; ++ip) {
        const int e = ip * DT + tid, l = e >> SH, c = e & (NC - 1);
        const bool ok = l < LC;
        const int lc = ok ? l : 0, ec = ok ? e : c, L = lc + ilev;
        const float *ys = d.yscale_lev + L * 5;
        const float pd = s_pd[lc], area = s_area[ec], qv = s_qv[ec], qn = s_qn[ec];
        const bool up = lc > 0, last = lc == LC - 1;
        const float flux_t_dp = ((last && d.dec21 ? 0.0f : s_fH[ec]) - (up ? s_fH[ec - NC] : 0.0f)) / pd * (-G / CP);
        const float flux_qv_dp = ((last ? 0.0f : s_fqv[ec]) - (up ? s_fqv[ec - NC] : 0.0f)) / pd * (-G);
        const float flux_qn_dp = ((last ? 0.0f : s_fqn[ec]) - (up ? s_fqn[ec - NC] : 0.0f)) / pd * (-G);
        const float sed_qn_dp = (s_sed[ec] - (up ? s_sed[ec - NC] : 0.0f)) / pd * (-G);
        float evap = (fmaxf(hC[ip][0], 0.0f) + 1e-6f) * s_pv[lc];
        float cond = hC[ip][1];
        float aa = fmaxf(hC[ip][2], 0.0f) * qn * ys[2];
        cond = fmaxf(cond, ((-(ys[2] * qn / 1200.0f) - flux_qn_dp) + aa) - sed_qn_dp);
        evap = fmaxf(evap, (-(ys[1] * qv / 1200.0f) - flux_qv_dp) + cond);
        aa = fmaxf(aa, ((flux_qn_dp + cond) + sed_qn_dp) - ys[2] * (-qn + 0.0006f) / 1200.0f);
        const float dqv = (flux_qv_dp - cond) + evap;
        const float dqn = ((flux_qn_dp + cond) - aa) + sed_qn_dp;
        const float *xd = x_denorm + ((size_t)b * PH_L + L) * nxd;
        // physRad graphs: one temperature per level (flux_t_dp is the same in every region) and latent heating from the
        // area-summed rates; otherwise per sub-column
        const float temp = xd[0] + ((d.gridT ? flux_t_dp : ph_sum<NC>(area * flux_t_dp)) / ys[0]) * 1200.0f;
        float liq = fminf(fmaxf((temp - 253.16f) * 0.05f, 0.0f), 1.0f);
        if (RAD && d.dec21) {  // per region: the ramp on the region's own temperature after the flux divergence, or the learned head
            liq = d.liq_off >= 0 ? 1.0f / (1.0f + expf(-HD[((size_t)(L - d.ltop) * B + b) * HDW + d.liq_off + c]))
                                 : fminf(fmaxf(((s_T[ec] + (flux_t_dp / ys[0]) * 1200.0f) - 253.16f) * 0.05f, 0.0f), 1.0f);
            if (ok) s_liq[e] = liq;
        }
        const float cond_h = d.gridT ? ph_sum<NC>(area * cond) : cond, evap_h = d.gridT ? ph_sum<NC>(area * evap) : evap;
        const float net = ((liq * LV + (1.0f - liq) * LS) * cond_h - evap_h * LV) * (1.0f / CP);
        const float dT_crm = flux_t_dp + net / ys[1] * ys[0];
        const float sT = ph_sum<NC>(area * dT_crm), sqv = ph_sum<NC>(area * dqv), sqn = ph_sum<NC>(area * dqn);
        const float sprec = ph_sum<NC>(area * (aa - evap));
        const float ssed = ph_sum<NC>(area * s_sed[ec]);
        if (RAD && ok) {                              // sub-column state after the step, as the radiation scheme sees it
            s_T[e] = fmaxf(s_T[e] + dT_crm * 1200.0f / ys[0], 0.0f);
            if (!d.cld_qn_old) s_qn[e] = fmaxf(qn + dqn * 1200.0f / ys[2], 0.0f);
            s_qv[e] = fmaxf(qv + dqv * 1200.0f / ys[1], 0.0f);
        }
        if (ok && c == 0) {
            float *o = out_lev + ((size_t)b * PH_L + L) * 5;
            if (RAD) {
                o[0] = sT;
                s_o01[L][0] = sT; s_o01[L][1] = sqv; s_o01[L][2] = sqn;
            } else {
                const float dT_rad = HD[((size_t)L * B + b) * HDW + HDW - 1];
                o[0] = ((l >= 2 ? s_out[l][0] : 0.0f) + sT) + dT_rad;
            }
            o[1] = sqv;
            o[2] = sqn;
            o[3] = l >= 2 ? s_out[l][3] : 0.0f;
            o[4] = l >= 2 ? s_out[l][4] : 0.0f;
            s_dprec[l] = pd * OOG * sprec;
            if (last) s_scal[8] = ssed;               // sedimentation reaching the surface
        }
    }
    // levels above the CRM top: only the radiative heating
    for (int L = tid; L < ilev; L += DT) {
        float *o = out_lev + ((size_t)b * PH_L + L) * 5;
        o[0] = RAD ? 0.0f : HD[((size_t)L * B + b) * HDW + HDW - 1];
        o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f; o[4] = 0.0f;
        if (RAD) { s_o01[L][0] = 0.0f; s_o01[L][1] = 0.0f; s_o01[L][2] = 0.0f; }
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
            os[2] = snowfrac * precc; os[3] = precc;
            if (!RAD) {
                os[0] = fmaxf(s_scal[0], 0.0f); os[1] = fmaxf(s_scal[1], 0.0f);
                for (int k = 2; k < 6; ++k) os[2 + k] = fmaxf(s_scal[k], 0.0f);
            }
            s_red[0] = stored;
        }
    }
    __syncthreads();
    for (int l = tid; l < LC; l += DT) mem_out[ph_mem_row(d, B, b, l) * (nm0 + 1) + nm0] = s_red[0];

    // ---- phase E (RAD): inputs of the radiation scheme -- the artefact's radiative_transfer up to its three MLPs ----
    if constexpr (RAD) {
        const float *aux = x_sfc + (size_t)b * d.naux;
        // One level per lane, THREE waves each doing a third of a level's row (the rows cost ~30 IEEE divisions, two fourth roots and a
        // logarithm each: 19 k cycles when one lane did a whole row -- a third of this kernel at a few hundred columns, measured with
        // cycle stamps -- while seven waves idled): wave 0 writes RS and XG[0:12], wave 1 XG[12:24], wave 2 the SW inputs XR; the
        // level's scalars are recomputed by each.  The other waves go straight to the cloud optics below.
        const int role = tid >> 6, L_e = tid & 63;
        if (role < 3 && L_e < PH_L) {
            const int L = L_e;
            const size_t row = (size_t)L * B + b;
            const float *xd = x_denorm + ((size_t)b * PH_L + L) * nxd;
            const float *ys = d.yscale_lev + L * 5;
            const float T_new = d.rad_T_old ? xd[0] : fmaxf(xd[0] + s_o01[L][0] / ys[0] * 1200.0f, 0.0f);
            const float qv_new = d.nx21 && !d.rad_qv_upd ? xd[nxd - 1] : fmaxf(xd[nxd - 1] + s_o01[L][1] / ys[1] * 1200.0f, 0.0f);
            const float vmr = (d.physrad || d.nx21 ? qv_new / (1.0f - qv_new) : qv_new) * 1.608079364f, fact = 1.0f / (1.0f + vmr), m_air = (vmr + 0.04698f) * fact;
            const float pd = sp * (d.hybi[L + 1] - d.hybi[L]) + (d.hyai[L + 1] - d.hyai[L]) * 100000.0f;
            const float col_dry = (pd * 10.0f * 6.02214076e23f * fact) / (m_air * 1000.0f * 100.0f * 9.80665f);
            const float play = d.hyam[L] * 100000.0f + sp * d.hybm[L], lp = logf(play), v4 = sqrtf(sqrtf(vmr));
            if (role < 2) {
                if (role == 0) { ro.RS[row * 2] = col_dry; ro.RS[row * 2 + 1] = T_new; }
                float f[PH_XG_K];
#pragma unroll
                for (int k = 0; k < PH_XG_K; ++k) f[k] = 0.0f;
                f[0] = T_new; f[1] = lp; f[2] = v4; f[3] = sqrtf(sqrtf(xd[12])); f[4] = 0.0003887f; f[5] = xd[13]; f[6] = xd[14];
                float *xg = ro.XG + row * PH_XG_K;
#pragma unroll
                for (int k = 0; k < PH_XG_K / 2; ++k) {
                    const int kk = k + (role ? PH_XG_K / 2 : 0);
                    xg[kk] = kk < 18 ? fmaxf(((role ? f[k + PH_XG_K / 2] : f[k]) - d.g_xmin[kk]) / d.g_range[kk], 0.0f) : 0.0f;
                }
            } else {
            float *xr = ro.XR + row * PH_XR_K;
            if (d.swg) {
                // physics_rad_e3sm generation: inputs of the SW gas-optics MLPs for the humidity of the two largest regions of the
                // level (above the CRM: the grid-mean value -- its FOURTH ROOT, as the serialised graph concatenates it)
                float v[2] = {v4, v4};
                if (L >= ilev) {
                    const float *ar = s_area + (L - ilev) * NC;
                    int i0 = 0, i1 = -1;
#pragma unroll
                    for (int c = 1; c < NC; ++c) if (ar[c] > ar[i0]) i0 = c;
#pragma unroll
                    for (int c = 0; c < NC; ++c) if (c != i0 && (i1 < 0 || ar[c] > ar[i1])) i1 = c;
                    const float q0 = fminf(s_qv[(L - ilev) * NC + i0], 0.05f), q1 = fminf(s_qv[(L - ilev) * NC + i1], 0.05f);
                    v[0] = q0 / (1.0f - q0) * 1.608079364f; v[1] = q1 / (1.0f - q1) * 1.608079364f;
                }
                const float *xmin = d.swg + SWG_XMIN, *xdiv = d.swg + SWG_XDIV;
                xr[0] = (T_new - xmin[0]) / xdiv[0]; xr[1] = (lp - xmin[1]) / xdiv[1];
                xr[3] = (sqrtf(sqrtf(xd[12])) - xmin[3]) / xdiv[3]; xr[4] = (0.0003887f - xmin[4]) / xdiv[4];
                xr[5] = (xd[14] - xmin[5]) / xdiv[5]; xr[6] = (xd[13] - xmin[6]) / xdiv[6];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float fj = 1.0f / (v[j] + 1.0f);
                    xr[j ? 7 : 2] = (sqrtf(sqrtf(v[j])) - xmin[2]) / xdiv[2];
                    xr[8 + j] = ((pd * 6.02214076e24f) * fj) / (((v[j] + 0.04698f) * fj) * 980665.0f);
                }
            } else {
            const float *xm = ro.x_main + ((size_t)b * PH_L + L) * d.nx;
            xr[0] = (lp - 0.00515f) / 11.59485f;
            xr[1] = (T_new - 160.0f) / 180.0f;
            xr[2] = sqrtf(sqrtf(qv_new * 1.608079364f)) / 0.497653f;
            const float qn_rad = d.rad_qn_upd ? fmaxf((xd[2] + xd[3]) + s_o01[L][2] / ys[2] * 1200.0f, 0.0f) : xd[2] + xd[3];
            xr[3] = 1.0f - expf(-qn_rad * d.lbd_qn[L]);
            xr[4] = xm[12]; xr[5] = xm[13]; xr[6] = xm[14];
            float rel = 0.0f, rei = 0.0f;
            if (L >= ilev) {
                rel = ph_reltab(T_new, aux[13] * d.xdiv_sca[13] + d.xmean_sca[13], aux[12] * d.xdiv_sca[12] + d.xmean_sca[12],
                                aux[15] * d.xdiv_sca[15] + d.xmean_sca[15]) / 13.5f;
                rei = ph_reitab(d.retab, T_new) / 250.0f;
            }
            xr[7] = rel; xr[8] = rei;
            const float *lat = HD + ((size_t)(L - d.ltop) * B + b) * HDW + PH_NHEAD * NC;
#pragma unroll
            for (int k = 0; k < nm0; ++k) xr[9 + k] = L >= ilev ? lat[k] : 0.0f;
            }   // SW head inputs
            }   // role 2
        }
        // cloud optical depth per (CRM level, g-point).  use_mcica graphs (mp_ncol 4): every g-point sees one sub-column,
        // sub-column j owning round-to-largest-remainder(area_j * 16) consecutive g-points (physics_rad.py:533); the
        // mp_ncol 16 graphs pair g-point g with sub-column g
        for (int e = tid; e < LC * PH_NG; e += DT) {
            const int l = e >> 4, g = e & 15, L = l + ilev;
            int sub = g, c0 = 0;
            if constexpr (NC != PH_NG) {
                // physRad graphs with MCICA (nreg 4): the g-points sample the CLOUDY regions 1.. only, fractions renormalised
                c0 = d.physrad ? 1 : 0;
                float psum = 1.0f;
                if (d.physrad) {
                    psum = 0.0f;
#pragma unroll
                    for (int c = 1; c < NC; ++c) psum += s_area[l * NC + c];
                }
                float rem[NC], cnt[NC], tot = 0.0f;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float a = s_area[l * NC + c];
                    const float p = (d.physrad ? a / psum : a) * (float)PH_NG;
                    cnt[c] = c < c0 ? 0.0f : floorf(p);
                    rem[c] = c < c0 ? -1.0f : p - cnt[c];               // (a skipped region never ranks)
                    tot += cnt[c];
                }
                const float deficit = (float)PH_NG - tot;
                float edge = 0.0f;
                sub = 0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    int rank = 0;
#pragma unroll
                    for (int i = 0; i < NC; ++i) rank += (rem[i] > rem[c] || (rem[i] == rem[c] && i < c)) ? 1 : 0;
                    if (c >= c0) {
                        edge += cnt[c] + ((float)rank < deficit ? 1.0f : 0.0f);
                        sub += edge <= (float)g ? 1 : 0;
                    }
                }
                sub = min(sub, NC - 1 - c0);
            }
            const float T_g = s_T[l * NC + c0 + sub], qn_g = s_qn[l * NC + c0 + sub];   // (the liquid-fraction head below is read at `sub`,
                                                                                      //  un-shifted, as the serialised graph gathers it)
            float liq = fminf(fmaxf((T_g - 253.16f) * 0.05f, 0.0f), 1.0f);
            if (d.liq_off >= 0) liq = 1.0f / (1.0f + expf(-HD[((size_t)(L - d.ltop) * B + b) * HDW + d.liq_off + sub]));
            if (d.dec21 && !d.cld_liq_upd) liq = s_liq[l * NC + sub];
            const float cwp = s_pd[l] / G * qn_g * 1000.0f, cwp_ice = (1.0f - liq) * cwp;
            const float ifr = cwp_ice / fmaxf(cwp, 1e-8f);
            const float *xd = x_denorm + ((size_t)b * PH_L + L) * nxd;
            const float *ys = d.yscale_lev + L * 5;
            const float T_new = d.rad_T_old ? xd[0] : fmaxf(xd[0] + s_o01[L][0] / ys[0] * 1200.0f, 0.0f);
            const float rei = fminf(fmaxf(ph_reitab(d.retab, T_new), 13.0f), 130.0f);
            float tau_cld = cwp * 0.090361f * (1.0f - ifr) + cwp * ifr * (1.0f / rei + 0.005f);
            if (d.cld_w) {          // learned optics: ReLU(Linear([(T_crm - 160) / 180, r_ice / 125, r_liq / 13.5, new memory (15 + stored water)]))
                const float *aux = x_sfc + (size_t)b * d.naux, *wr = d.cld_w + g * 19;
                const float rel = ph_reltab(T_new, aux[13] * d.xdiv_sca[13] + d.xmean_sca[13], aux[12] * d.xdiv_sca[12] + d.xmean_sca[12],
                                            aux[15] * d.xdiv_sca[15] + d.xmean_sca[15]);
                const float *lat = HD + ((size_t)(L - d.ltop) * B + b) * HDW + PH_NHEAD * NC;
                float a = d.cld_b[g];
                a = fmaf(wr[0], (s_T[l * NC] - 160.0f) / 180.0f, a);
                a = fmaf(wr[1], ph_reitab(d.retab, T_new) / 125.0f, a);
                a = fmaf(wr[2], rel / 13.5f, a);
#pragma unroll
                for (int k = 0; k < nm0; ++k) a = fmaf(wr[3 + k], lat[k], a);
                a = fmaf(wr[3 + nm0], s_red[0], a);
                tau_cld = cwp * fmaxf(a, 0.0f);
            }
            ro.CL[((size_t)l * B + b) * PH_NG + g] = tau_cld;
            if (d.swg && d.cld_sw_w) {   // learned SW cloud optics on the LW scheme's inputs: extinction per unit path, albedo, asymmetry of g-point g
                const float *aux = x_sfc + (size_t)b * d.naux;
                const float rel = ph_reltab(T_new, aux[13] * d.xdiv_sca[13] + d.xmean_sca[13], aux[12] * d.xdiv_sca[12] + d.xmean_sca[12],
                                            aux[15] * d.xdiv_sca[15] + d.xmean_sca[15]);
                const float *lat = HD + ((size_t)(L - d.ltop) * B + b) * HDW + PH_NHEAD * NC;
                const float x0 = (s_T[l * NC] - 160.0f) / 180.0f, x1 = ph_reitab(d.retab, T_new) / 125.0f, x2 = rel / 13.5f;
                float o3[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float *wr = d.cld_sw_w + (q * PH_NG + g) * 19;
                    float a = d.cld_sw_b[q * PH_NG + g];
                    a = fmaf(wr[0], x0, a); a = fmaf(wr[1], x1, a); a = fmaf(wr[2], x2, a);
#pragma unroll
                    for (int k = 0; k < nm0; ++k) a = fmaf(wr[3 + k], lat[k], a);
                    o3[q] = fmaf(wr[3 + nm0], s_red[0], a);
                }
                const float c_tau = cwp * fmaxf(o3[0], 0.0f);
                float *cs = ro.CS + ((size_t)l * B + b) * 48 + g;
                cs[0] = c_tau;
                cs[16] = c_tau / (1.0f + expf(-o3[1]));
                cs[32] = 1.0f / (1.0f + expf(-o3[2]));
            } else if (d.swg) {     // Slingo liquid / Ebert-Curry ice SW optics of region g (physics_rad_e3sm.py:98, :265)
                const float *aux = x_sfc + (size_t)b * d.naux, *t = (d.nx21 ? d.cldtab : d.swg + SWG_CLD) + g;
                const float rl0 = ph_reltab(T_new, aux[13] * d.xdiv_sca[13] + d.xmean_sca[13], aux[12] * d.xdiv_sca[12] + d.xmean_sca[12],
                                            aux[15] * d.xdiv_sca[15] + d.xmean_sca[15]);
                const float rl = fminf(fmaxf(rl0, 4.2f), 16.0f);
                const float ri = (d.nx21 && !d.ice_re) ? fminf(fmaxf(rl0, 13.0f), 130.0f) : rei;     // first nx21 exports: the ice optics see the LIQUID radius, as serialised
                float kl, sl, sgl, ki, si, sgi;
                if (d.cld_band) {      // four bands, then the learned band -> g-point matrix on k, k ssa, k ssa g
                    const float *tb = d.cldtab, *Mb = d.cldtab + 48 + g;
                    kl = sl = sgl = ki = si = sgi = 0.0f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float m = Mb[16 * q];
                        const float k1 = tb[q] + tb[4 + q] / rl, w1 = fminf((1.0f - tb[8 + q]) - rl * tb[12 + q], 0.999999f), g1 = tb[16 + q] + rl * tb[20 + q];
                        const float k2 = tb[24 + q] + tb[28 + q] / ri, w2 = fminf((1.0f - tb[32 + q]) - ri * tb[36 + q], 0.999999f), g2 = tb[40 + q] + ri * tb[44 + q];
                        kl = fmaf(k1, m, kl); sl = fmaf(k1 * w1, m, sl); sgl = fmaf((k1 * w1) * g1, m, sgl);
                        ki = fmaf(k2, m, ki); si = fmaf(k2 * w2, m, si); sgi = fmaf((k2 * w2) * g2, m, sgi);
                    }
                } else {
                    const float wl = fminf((1.0f - t[32]) - rl * t[48], 0.999999f), gl = t[64] + rl * t[80];
                    const float wi = fminf((1.0f - t[128]) - ri * t[144], 0.999999f), gi = t[160] + ri * t[176];
                    kl = t[0] + t[16] / rl; ki = t[96] + t[112] / ri;
                    sl = kl * wl; si = ki * wi; sgl = sl * gl; sgi = si * gi;
                }
                const float cwp_liq = liq * cwp;
                const float c_sca = cwp_liq * sl + cwp_ice * si;
                float *cs = ro.CS + ((size_t)l * B + b) * 48 + g;
                cs[0] = cwp_ice * ki + cwp_liq * kl;
                cs[16] = c_sca;
                cs[32] = (cwp_liq * sgl + cwp_ice * sgi) / (c_sca + 1e-7f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
struct PhysHostW {           // host pointers of one state_dict, by role
    const float *hyam, *hybm, *hyai, *hybi, *ysl, *yss, *xds, *xms, *init_w, *init_b, *s1_w, *s1_b;
    const float *r1_ih, *r1_hh, *r1_bih, *r1_bhh, *r2_ih, *r2_hh, *r2_bih, *r2_bhh;
    const float *lat_w, *lat_b, *out_w, *out_b, *sfo_w, *sfo_b, *rad_w, *rad_b, *rel_w, *rel_b;
    const float *const *heads;    // 11 x (weight (mp_ncol, nh), bias) in head-GEMM column order
    // radiation scheme
    const float *lbd_qn, *ys_rad, *solar_w, *g_xmin, *g_xmax /* or the range itself, see lw_dn */, *g_ymean, *g_ystd;
    const float *g_w1, *g_b1, *g_w2, *g_b2, *g_w3, *g_b3, *r1_w, *r1_b, *r2_w, *r2_b, *sw1_w, *sw1_b, *sw2_w, *sw2_b;
    const float *liq_w, *liq_b;              // mlp_liq_frac_crm (mp_ncol, nh), optional
    const float *s3_ih, *s3_zh, *s3_enc;     // rnn3 = MyStochasticGRULayer5(nh, nh) without bias, optional
    const float *cld_w, *cld_b;              // cloud_optics_lw (16, 19), optional
    const float *swg;                        // CSA_PHYS_SW_GAS block (SWG_FLOATS), optional
    const float *cld_sw_w, *cld_sw_b;        // composed learned SW cloud optics (48, 19), (48), optional
    // nx21 generation (csa_phys_wrapped_create): SWX_* block, cloud-optics table (12, 16), [n_ir, n_mix, mix_near, mix_vis]
    const float *swx = nullptr, *cldtab = nullptr, *misc = nullptr;
    int rad_qv_upd = 0, nx21 = 0;
};

// ice effective radius (micron) against temperature, 137 K ... : E3SM's table as listed in rnn/models/physics_rad_e3sm.py:13-59
static const float kRetab[PH_NRETAB] = {
    0.05f, 0.05f, 0.05f, 0.05f, 0.05f, 0.05f, 0.055f, 0.06f, 0.07f, 0.08f, 0.09f, 0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f, 0.9f, 1.0f,
    1.1f, 1.2f, 1.3f, 1.4f, 1.5f, 1.6f, 1.8f, 2.0f, 2.2f, 2.4f, 2.6f, 2.8f, 3.0f, 3.2f, 3.5f, 3.8f, 4.1f, 4.4f, 4.7f, 5.0f, 5.3f, 5.6f, 5.92779f,
    6.26422f, 6.61973f, 6.99539f, 7.39234f, 7.81177f, 8.25496f, 8.72323f, 9.218f, 9.74075f, 10.293f, 10.8765f, 11.4929f, 12.144f,
    12.8317f, 13.5581f, 14.2319f, 15.0351f, 15.8799f, 16.7674f, 17.6986f, 18.6744f, 19.6955f, 20.7623f, 21.8757f, 23.0364f, 24.2452f,
    25.5034f, 26.8125f, 27.7895f, 28.645f, 29.4167f, 30.1088f, 30.7306f, 31.2943f, 31.8151f, 32.3077f, 32.787f, 33.2657f, 33.754f,
    34.2601f, 34.7892f, 35.3442f, 35.9255f, 36.5316f, 37.1602f, 37.8078f, 38.472f, 39.1508f, 39.8442f, 40.5552f, 41.2912f, 42.0635f,
    42.8876f, 43.7863f, 44.7853f, 45.917f, 47.2165f, 48.7221f, 50.471f, 52.498f, 54.8315f, 57.4898f, 60.4785f, 63.7898f, 65.5604f,
    71.2885f, 75.4113f, 79.7368f, 84.2351f, 88.8833f, 93.6658f, 98.5739f, 103.603f, 108.752f, 114.025f, 119.424f, 124.954f, 130.63f,
    136.457f, 142.446f, 148.608f, 154.956f, 161.503f, 168.262f, 175.248f, 182.473f, 189.952f, 197.699f, 205.728f, 214.055f, 222.694f,
    231.661f, 240.971f, 250.639f};

static int phys_build(int nx, int nfeat, int naux, int nx_sfc, int sfc_cut, int nh, int ilev_crm, int mp_ncol, int nh_mem0, int rad, int physrad,
                      int lw_dn, const PhysHostW &w, int max_batch, csa_phys **out)
{
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
    PhysDev &d = h->d;
    d.nx = nx; d.nfeat = nfeat; d.naux = naux; d.nx_sfc = nx_sfc; d.sfc_cut = sfc_cut; d.sfc_skip = naux - nx_sfc;
    d.nh = nh; d.ilev = ilev_crm; d.nm0 = nh_mem0; d.Lc = PH_L - ilev_crm;
    d.ltop = rad ? ilev_crm : 0; d.Lr = PH_L - d.ltop;
    d.ncol = mp_ncol; d.rad = rad; d.physrad = physrad; d.memlm = physrad; d.gridT = d.clear0 = physrad;
    d.liq_off = w.liq_w ? PH_NHEAD * mp_ncol + 16 : -1;
    d.hdw = rad ? ((PH_NHEAD * mp_ncol + nh_mem0 + (w.liq_w ? 1 + mp_ncol : 0) + 3) / 4) * 4 : PH_NHEAD * mp_ncol + nh_mem0 + 1;
    d.hyam = up(w.hyam, 60); d.hybm = up(w.hybm, 60); d.hyai = up(w.hyai, 61); d.hybi = up(w.hybi, 61);
    d.yscale_lev = up(w.ysl, 60 * 5); d.yscale_sca = up(w.yss, 8);
    d.xdiv_sca0 = w.xds[0]; d.xmean_sca0 = w.xms[0];
    { auto t = transposed(w.init_w, nh, nfeat + 1); d.init_wt = up(t.data(), t.size()); }
    d.init_b = up(w.init_b, nh);
    { auto t = transposed(w.s1_w, nh, nx_sfc); d.s1_wt = up(t.data(), t.size()); }
    d.s1_b = up(w.s1_b, nh);
    d.out_w = up(w.out_w, 5 * nh_mem0); d.out_b = up(w.out_b, 5);
    if (!rad) { d.sfo_w = up(w.sfo_w, 6 * nh); d.sfo_b = up(w.sfo_b, 6); }
    d.rel_w = up(w.rel_w, nh); d.rel_b = up(w.rel_b, 1);
    // GRU layers: W_ih rows to unit-major [r, z, n]; rnn1's K = nh + 15 padded to nh + 16 with a zero column
    {
        const int Kin = nh + nh_mem0, K1 = nh + 16;
        std::vector<float> wpad((size_t)3 * nh * K1, 0.0f), wp, bp, bhn;
        for (int r = 0; r < 3 * nh; ++r) memcpy(&wpad[(size_t)r * K1], &w.r1_ih[(size_t)r * Kin], sizeof(float) * Kin);
        pack_ih(0, nh, K1, wpad.data(), w.r1_bih, w.r1_bhh, wp, bp, bhn);
        h->wih1 = up(wp.data(), wp.size()); h->bias1 = up(bp.data(), bp.size()); h->bhn1 = up(bhn.data(), bhn.size());
        pack_ih(0, nh, nh, w.r2_ih, w.r2_bih, w.r2_bhh, wp, bp, bhn);
        h->wih2 = up(wp.data(), wp.size()); h->bias2 = up(bp.data(), bp.size()); h->bhn2 = up(bhn.data(), bhn.size());
        std::vector<float> pk(rec_packed_floats(0, nh));
        rec_pack_weights(0, nh, w.r1_hh, pk.data()); h->whh1p = up(pk.data(), pk.size());
        rec_pack_weights(0, nh, w.r2_hh, pk.data()); h->whh2p = up(pk.data(), pk.size());
        gru2_pack_weights(nh, w.r1_hh, pk.data()); h->whh1g = up(pk.data(), pk.size());   // two-column kernel (B > 256)
        gru2_pack_weights(nh, w.r2_hh, pk.data()); h->whh2g = up(pk.data(), pk.size());
        {   // matrix-pipe kernel (four columns per workgroup, from 544 columns)
            std::vector<float> pm((size_t)4 * nh * nh);
            gru4m_pack_weights(nh, w.r1_hh, pm.data()); h->whh1m = up(pm.data(), pm.size());
            gru4m_pack_weights(nh, w.r2_hh, pm.data()); h->whh2m = up(pm.data(), pm.size());
        }
    }
    // head GEMM: 11 decoder heads (mp_ncol rows each), mlp_latent (15 rows), then mlp_output_rad (1 row) or zero padding
    {
        const int HDW = d.hdw;
        std::vector<float> wh((size_t)HDW * nh, 0.0f), bh(HDW, 0.0f);
        for (int k = 0; k < PH_NHEAD; ++k) {
            float *wk = &wh[(size_t)k * mp_ncol * nh], *bk = &bh[k * mp_ncol];
            if (physrad && k == H_T) continue;                            // no sub-grid temperature: the head stays zero
            if (physrad && (k == H_QN || k == H_COND)) {                  // mp_ncol - 1 cloudy regions; row 0 (clear sky) stays zero
                memcpy(wk + nh, w.heads[2 * k], sizeof(float) * (mp_ncol - 1) * nh);
                memcpy(bk + 1, w.heads[2 * k + 1], sizeof(float) * (mp_ncol - 1));
            } else if (physrad && k == H_EDDY) {                          // one diffusivity per level: the same row for every region
                for (int c = 0; c < mp_ncol; ++c) { memcpy(wk + (size_t)c * nh, w.heads[2 * k], sizeof(float) * nh); bk[c] = w.heads[2 * k + 1][0]; }
            } else {
                memcpy(wk, w.heads[2 * k], sizeof(float) * mp_ncol * nh);
                memcpy(bk, w.heads[2 * k + 1], sizeof(float) * mp_ncol);
            }
        }
        memcpy(&wh[(size_t)PH_NHEAD * mp_ncol * nh], w.lat_w, sizeof(float) * nh_mem0 * nh);
        memcpy(&bh[PH_NHEAD * mp_ncol], w.lat_b, sizeof(float) * nh_mem0);
        if (!rad) {
            memcpy(&wh[(size_t)(HDW - 1) * nh], w.rad_w, sizeof(float) * nh);
            bh[HDW - 1] = w.rad_b[0];
        }
        if (w.liq_w) {
            memcpy(&wh[(size_t)d.liq_off * nh], w.liq_w, sizeof(float) * mp_ncol * nh);
            memcpy(&bh[d.liq_off], w.liq_b, sizeof(float) * mp_ncol);
        }
        h->whead = up(wh.data(), wh.size()); h->bhead = up(bh.data(), bh.size());
    }
    const size_t rows = (size_t)d.Lr * max_batch;
    h->X1 = up(nullptr, rows * (nh + 16)); h->P = up(nullptr, rows * 4 * nh); h->H1 = up(nullptr, rows * nh);
    h->H2 = up(nullptr, rows * nh); h->hx = up(nullptr, (size_t)max_batch * nh); h->HD = up(nullptr, rows * d.hdw);
    if (rad) {
        d.xmean_sca = up(w.xms, naux); d.xdiv_sca = up(w.xds, naux); d.lbd_qn = w.lbd_qn ? up(w.lbd_qn, 60) : nullptr;
        {   // normalisation range of the gas-optics inputs: xmax - xmin (float, as the reference subtracts) or the stored `xdiv`
            float rg[18];
            for (int k = 0; k < 18; ++k) rg[k] = lw_dn ? w.g_xmax[k] : w.g_xmax[k] - w.g_xmin[k];
            d.g_range = up(rg, 18);
        }
        // (num94634, the SW gas-optics graph with the Slingo / Ebert-Curry cloud optics, carries `xdiv` too but views the upward source twice again)
        d.lw_dn = lw_dn && !(w.swg && !w.cld_sw_w);
        d.n_ir = 11; d.n_mix = 13; d.mix_near = 0.5f; d.mix_vis = 0.5f;      // round(0.7143 * 16), round(0.7946 * 16)
        d.cld_w = w.cld_w ? up(w.cld_w, PH_NG * 19) : nullptr;
        d.cld_b = w.cld_w ? up(w.cld_b, PH_NG) : nullptr;
        d.g_xmin = up(w.g_xmin, 18); d.g_ymean = up(w.g_ymean, 128); d.g_ystd = up(w.g_ystd, 128);
        d.ys_rad = up(w.ys_rad, 6); d.retab = up(kRetab, PH_NRETAB);
        {   // incoming spectral weights: softmax of the squared learned weights (physRad graphs: un-squared), float arithmetic
            float sq[PH_NG], m = -3.0e38f, sum = 0.0f, e[PH_NG];
            for (int g = 0; g < PH_NG; ++g) { sq[g] = physrad ? w.solar_w[g] : w.solar_w[g] * w.solar_w[g]; m = sq[g] > m ? sq[g] : m; }
            for (int g = 0; g < PH_NG; ++g) { e[g] = expf(sq[g] - m); sum += e[g]; }
            for (int g = 0; g < PH_NG; ++g) e[g] /= sum;
            d.toa_spec = w.nx21 ? up(w.solar_w, PH_NG) : up(e, PH_NG);     // (frozen exports: the weights arrive as the folded constant)
        }
        auto padK = [&](const float *src, int n, int k, int kp) {
            std::vector<float> t((size_t)n * kp, 0.0f);
            for (int r = 0; r < n; ++r) memcpy(&t[(size_t)r * kp], &src[(size_t)r * k], sizeof(float) * k);
            return up(t.data(), t.size());
        };
        h->g_w1 = padK(w.g_w1, 64, 18, PH_XG_K); h->g_b1 = up(w.g_b1, 64);
        h->g_w2 = up(w.g_w2, 64 * 64); h->g_b2 = up(w.g_b2, 64);
        h->g_w3 = up(w.g_w3, 256 * 64); h->g_b3 = up(w.g_b3, 256);
        h->r1_w = up(w.r1_w, 16 * 128); h->r1_b = up(w.r1_b, 16);
        h->r2_w = up(w.r2_w, 16 * 128); h->r2_b = up(w.r2_b, 16);
        if (w.nx21) {         // the nx21 generation of the frozen exports
            const int bits = (int)w.misc[7];
            d.sw_ngk = (int)w.misc[4]; d.ice_re = (int)w.misc[5]; d.cld_band = (int)w.misc[6];
            d.sfc_sw_down = bits & 1; d.cld_liq_upd = (bits >> 1) & 1; d.rad_qn_upd = (bits >> 2) & 1;
            d.gridT = (bits >> 3) & 1; d.clear0 = (bits >> 4) & 1; d.cld_qn_old = (bits >> 5) & 1; d.dec21 = !d.gridT;
            d.rad_T_old = (bits >> 6) & 1; h->rnn3_last_mul = (bits >> 7) & 1;
            if (w.swx || w.swg) {      // SW gas-optics MLPs + Slingo / Ebert-Curry cloud optics
                d.sw_e3sm = w.swg != nullptr;             // (the unfrozen physics_rad_e3sm form of the gas optics under the wrapper)
                d.swg = d.sw_e3sm ? up(w.swg, SWG_FLOATS) : up(w.swx, SWX_FLOATS);
                d.cldtab = up(w.cldtab, d.cld_band ? 12 * 4 + 4 * PH_NG : 12 * PH_NG);
                h->CS = up(nullptr, (size_t)d.Lc * max_batch * 48);
            } else {          // earlier sub-generation: the SW head MLP (24 -> 32 -> 3 x 16) of the unfrozen num4050 family
                h->s1_w = up(w.sw1_w, 32 * PH_XR_K); h->s1_b = up(w.sw1_b, 32);
                h->s2_w = up(w.sw2_w, 48 * 32); h->s2_b = up(w.sw2_b, 48);
            }
            d.nx21 = 1; d.memlm = 1; d.lw_dn = 1; d.rad_qv_upd = w.rad_qv_upd;
            d.n_ir = (int)w.misc[0]; d.n_mix = (int)w.misc[1]; d.mix_near = w.misc[2]; d.mix_vis = w.misc[3];
        } else if (w.swg) {
            d.swg = up(w.swg, SWG_FLOATS);
            h->CS = up(nullptr, (size_t)d.Lc * max_batch * 48);
            if (w.cld_sw_w) { d.cld_sw_w = up(w.cld_sw_w, 48 * 19); d.cld_sw_b = up(w.cld_sw_b, 48); }
        } else {
            h->s1_w = up(w.sw1_w, 32 * PH_XR_K); h->s1_b = up(w.sw1_b, 32);
            h->s2_w = up(w.sw2_w, 48 * 32); h->s2_b = up(w.sw2_b, 48);
        }
        const size_t M = (size_t)PH_L * max_batch;
        h->XG = up(nullptr, M * PH_XG_K); h->XR = up(nullptr, M * PH_XR_K); h->RS = up(nullptr, M * 2);
        h->CL = up(nullptr, (size_t)d.Lc * max_batch * PH_NG);
        h->TP = up(nullptr, M * 32); h->S2 = up(nullptr, M * 48);
    }
    if (w.s3_ih && rc == CSA_OK) {
        rc = csa_stoch_gru5_create(nh, nh, w.s3_ih, w.s3_zh, w.s3_enc, nullptr, nullptr, (int)rows, &h->rnn3);
        h->H3 = up(nullptr, rows * nh); h->H2p = up(nullptr, rows * nh);
    }
    if (rc != CSA_OK) { if (h->rnn3) csa_stoch_destroy(h->rnn3); for (void *p : h->owned) (void)hipFree(p); delete h; return rc; }
    *out = h;
    return CSA_OK;
}

extern "C" int csa_phys_create(int nx, int nx_sfc, int nh, int ilev_crm, int mp_ncol, int nh_mem0,
                               const float *const *w /* see include/climsim_amd.h for the order */, int max_batch, csa_phys **out)
{
    if (!w || !out || max_batch <= 0) { csa_set_error_msg("csa_phys_create: bad argument"); return CSA_ERR_ARG; }
    if (nh != 128 || mp_ncol != 16 || nh_mem0 != 15 || ilev_crm != 10 || nx + 1 > 32 || nx_sfc > 64) {
        csa_set_error_msg("csa_phys_create: built for the shipped physRNN-Hidden geometry (GRU 128/128, mp_ncol 16, 15+1 memory channels, ilev_crm 10)");
        return CSA_ERR_UNSUPPORTED;
    }
    PhysHostW v{};
    const float *const *p = w;
    v.hyam = *p++; v.hybm = *p++; v.hyai = *p++; v.hybi = *p++; v.ysl = *p++; v.yss = *p++; v.xds = *p++; v.xms = *p++;
    v.init_w = *p++; v.init_b = *p++; v.s1_w = *p++; v.s1_b = *p++;
    v.r1_ih = *p++; v.r1_hh = *p++; v.r1_bih = *p++; v.r1_bhh = *p++; v.r2_ih = *p++; v.r2_hh = *p++; v.r2_bih = *p++; v.r2_bhh = *p++;
    v.lat_w = *p++; v.lat_b = *p++; v.out_w = *p++; v.out_b = *p++; v.sfo_w = *p++; v.sfo_b = *p++; v.rad_w = *p++; v.rad_b = *p++;
    v.rel_w = *p++; v.rel_b = *p++;
    v.heads = p;
    int rc = phys_build(nx, nx, nx_sfc, nx_sfc, nx_sfc, nh, ilev_crm, mp_ncol, nh_mem0, 0, 0, 0, v, max_batch, out);
    if (rc != CSA_OK) return rc;
    // the trainable tensors in state_dict order (csa_phys_train_param_info): kept on the host until csa_phys_train_enable
    std::vector<float> &hp = (*out)->host_params;
    auto app = [&](const float *src, size_t n) { hp.insert(hp.end(), src, src + n); };
    app(v.init_w, (size_t)nh * (nx + 1)); app(v.init_b, nh); app(v.s1_w, (size_t)nh * nx_sfc); app(v.s1_b, nh);
    app(v.r1_ih, (size_t)3 * nh * (nh + nh_mem0)); app(v.r1_hh, (size_t)3 * nh * nh); app(v.r1_bih, 3 * nh); app(v.r1_bhh, 3 * nh);
    app(v.r2_ih, (size_t)3 * nh * nh); app(v.r2_hh, (size_t)3 * nh * nh); app(v.r2_bih, 3 * nh); app(v.r2_bhh, 3 * nh);
    app(v.lat_w, (size_t)nh_mem0 * nh); app(v.lat_b, nh_mem0); app(v.out_w, 5 * nh_mem0); app(v.out_b, 5);
    app(v.sfo_w, 6 * nh); app(v.sfo_b, 6); app(v.rad_w, nh); app(v.rad_b, 1); app(v.rel_w, nh); app(v.rel_b, 1);
    for (int k = 0; k < PH_NHEAD; ++k) { app(v.heads[2 * k], (size_t)mp_ncol * nh); app(v.heads[2 * k + 1], mp_ncol); }
    return CSA_OK;
}

// The radiation graphs (num4050): see include/climsim_amd.h for the pointer order
extern "C" int csa_phys_rad_create(int nx, int naux, int nh, int ilev_crm, int mp_ncol, int nh_mem0, int ng, int flags,
                                   const float *const *w, int max_batch, csa_phys **out)
{
    if (!w || !out || max_batch <= 0 || (flags & ~255)) { csa_set_error_msg("csa_phys_rad_create: bad argument"); return CSA_ERR_ARG; }
    const bool mcica = flags & CSA_PHYS_MCICA, physrad = flags & CSA_PHYS_PHYSRAD;
    if (physrad && !(flags & (CSA_PHYS_LIQ_FRAC_HEAD | CSA_PHYS_CLOUD_OPTICS_LW))) {
        csa_set_error_msg("csa_phys_rad_create: the physRad graphs come with the liquid-fraction head or the learned cloud optics");
        return CSA_ERR_UNSUPPORTED;
    }
    if ((flags & CSA_PHYS_CLOUD_OPTICS_LW) && (!physrad || mcica)) {
        csa_set_error_msg("csa_phys_rad_create: the learned cloud optics belong to the 16-region physRad graphs");
        return CSA_ERR_UNSUPPORTED;
    }
    if ((nh != 128 && nh != 112 && nh != 96) || mp_ncol != (mcica ? 4 : 16) || nh_mem0 != 15 || ilev_crm != 10 || (nx != 21 && nx != 16) || naux != 19 ||
        ng != PH_NG) {
        csa_set_error_msg("csa_phys_rad_create: built for the shipped geometries (21 or 16 level inputs, 19 surface inputs, GRU 128 / 112 / 96 over 50 "
                          "levels, 15+1 memory channels, 16 g-points; mp_ncol 4 with MCICA sampling or mp_ncol 16 without)");
        return CSA_ERR_UNSUPPORTED;
    }
    PhysHostW v{};
    const float *const *p = w;
    v.hyam = *p++; v.hybm = *p++; v.hyai = *p++; v.hybi = *p++; v.ysl = *p++; v.yss = *p++; v.xds = *p++; v.xms = *p++;
    v.init_w = *p++; v.init_b = *p++; v.s1_w = *p++; v.s1_b = *p++;
    v.r1_ih = *p++; v.r1_hh = *p++; v.r1_bih = *p++; v.r1_bhh = *p++; v.r2_ih = *p++; v.r2_hh = *p++; v.r2_bih = *p++; v.r2_bhh = *p++;
    v.lat_w = *p++; v.lat_b = *p++; v.out_w = *p++; v.out_b = *p++; v.rel_w = *p++; v.rel_b = *p++;
    v.heads = p; p += 2 * PH_NHEAD;
    v.lbd_qn = *p++; v.ys_rad = *p++; v.solar_w = *p++; v.g_xmin = *p++; v.g_xmax = *p++; v.g_ymean = *p++; v.g_ystd = *p++;
    v.g_w1 = *p++; v.g_b1 = *p++; v.g_w2 = *p++; v.g_b2 = *p++; v.g_w3 = *p++; v.g_b3 = *p++;
    v.r1_w = *p++; v.r1_b = *p++; v.r2_w = *p++; v.r2_b = *p++;
    const float *const *const sw_head = p;           // the four SW-head slots: unused (may be null) with the SW gas-optics models
    v.sw1_w = *p++; v.sw1_b = *p++; v.sw2_w = *p++; v.sw2_b = *p++;
    if (flags & CSA_PHYS_LIQ_FRAC_HEAD) { v.liq_w = *p++; v.liq_b = *p++; }
    if (flags & CSA_PHYS_STOCHASTIC) { v.s3_ih = *p++; v.s3_zh = *p++; v.s3_enc = *p++; }
    if (flags & CSA_PHYS_CLOUD_OPTICS_LW) { v.cld_w = *p++; v.cld_b = *p++; }
    const float *const *swh = nullptr;
    if (flags & CSA_PHYS_SW_GAS) {
        const bool learned = (flags & CSA_PHYS_CLOUD_OPTICS_LW) && (flags & CSA_PHYS_CLOUD_OPTICS_SW);
        const bool tabled = (flags & CSA_PHYS_LIQ_FRAC_HEAD) && !(flags & (CSA_PHYS_CLOUD_OPTICS_LW | CSA_PHYS_CLOUD_OPTICS_SW));
        if (!physrad || mcica || !(learned || tabled)) {
            csa_set_error_msg("csa_phys_rad_create: the SW gas-optics models belong to the 16-region physRad graphs, with the liquid-fraction head "
                              "(Slingo / Ebert-Curry cloud optics) or with both learned cloud-optics layers");
            return CSA_ERR_UNSUPPORTED;
        }
        v.swg = *p++;
        swh = sw_head;
        if (flags & CSA_PHYS_CLOUD_OPTICS_SW) { v.cld_sw_w = *p++; v.cld_sw_b = *p++; }
    } else if (flags & CSA_PHYS_CLOUD_OPTICS_SW) {
        csa_set_error_msg("csa_phys_rad_create: the learned SW cloud optics come with the SW gas-optics models");
        return CSA_ERR_UNSUPPORTED;
    }
    for (const float *const *q = w; q != p; ++q)
        if (!*q && !(physrad && (q == v.heads + 2 * H_T || q == v.heads + 2 * H_T + 1)) && !(swh && q >= swh && q < swh + 4)) {      // (no mlp_t_crm in the physRad graphs)
            csa_set_error_msg("csa_phys_rad_create: null weight pointer");
            return CSA_ERR_ARG;
        }
    // mlp_initial sees x_main[:, :, 0:nx-3] and the layer pressure; mlp_surface1 sees aux 0:6 and 11:naux
    return phys_build(nx, nx - 3, naux, naux - 5, 6, nh, ilev_crm, mp_ncol, nh_mem0, 1, physrad ? 1 : 0, (flags & CSA_PHYS_LATER_EXPORT) ? 1 : 0, v, max_batch, out);
}

// ---- the frozen `*_wrapped` exports: rnn/utils.py::model_wrapper (:72-295) inlined around the nx21 generation of the model ----------
// Wrapper pre-processing (:134-217, :262-272): q = RH * Rd esat(T) / (Rv p) appended as 21st level input, snow / ice sentinel -> -1,
// 1 - exp(-lambda q) on the two cloud inputs, (x - mean) / div, NaN and Inf -> 0.  One thread per (column, level); the raw row with
// q (what the model calls inputs_denorm) and the normalised row are both written.
__device__ __forceinline__ float ph_horner9(const float *a, float x)
{
    float o = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) o = o * x + a[i];       // the reference's loop: out = out * x + c (unfused, as torch evaluates it)
    return o;
}
__global__ __launch_bounds__(256) void phys_wrap_pre_kernel(PhysDev d, int B, const float *__restrict__ x_main0, const float *__restrict__ x_sfc0,
                                                            const float *__restrict__ xmean, const float *__restrict__ xdiv,
                                                            const float *__restrict__ lqc, const float *__restrict__ lqi,
                                                            float *__restrict__ XM, float *__restrict__ XS, float *__restrict__ XD)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B * PH_L) {
        const int b = i / PH_L, L = i - b * PH_L;
        const float *x = x_main0 + (size_t)i * 20;
        const float a_liq[9] = {-0.976195544e-15f, -0.952447341e-13f, 0.640689451e-10f, 0.206739458e-7f, 0.302950461e-5f, 0.264847430e-3f,
                                0.142986287e-1f, 0.443987641f, 6.11239921f};
        const float a_ice[9] = {0.252751365e-14f, 0.146898966e-11f, 0.385852041e-9f, 0.602588177e-7f, 0.615021634e-5f, 0.420895665e-3f,
                                0.188439774e-1f, 0.503160820f, 6.11147274f};
        const float temp = x[0], rh = x[1];
        const float pres = d.hyam[L] * 100000.0f + x_sfc0[(size_t)b * d.naux] * d.hybm[L];
        const float omega = fminf(fmaxf((temp - 253.16f) / 20.000000000000028f, 0.0f), 1.0f);
        const float eliq = ph_horner9(a_liq, fmaxf(temp - 273.16f, -80.0f)) * 100.0f;
        const float b2 = ph_horner9(a_ice, temp - 273.16f) * 100.0f;
        const float tmp = fmaxf(temp - 273.16f, -100.0f);
        const float b3 = (0.00763685f + tmp * (0.000151069f + tmp * 7.48215e-07f)) * 100.0f;
        const float eice = temp > 273.15f ? eliq : (temp > 185.0f ? b2 : b3);
        const float esat = omega * eliq + (1.0f - omega) * eice;
        const float q = rh * ((esat * 287.0f) / (pres * 461.0f));
        float *xd = XD + (size_t)i * 21, *xm = XM + (size_t)i * 21;
#pragma unroll
        for (int k = 0; k < 21; ++k) {
            const float raw = k < 20 ? x[k] : q;
            xd[k] = raw;
            float v = raw;
            if (k == 2) v = 1.0f - expf(-raw * lqc[L]);
            if (k == 3) v = 1.0f - expf(-raw * lqi[L]);
            v = (v - xmean[L * 21 + k]) / xdiv[L * 21 + k];
            xm[k] = (v != v || fabsf(v) > 3.402823466e38f) ? 0.0f : v;
        }
    }
    if (i < B * d.naux) {
        const int k = i % d.naux;
        const float v = x_sfc0[i];
        XS[i] = ((v >= 1.0e10f ? -1.0f : v) - d.xmean_sca[k]) / d.xdiv_sca[k];
    }
}

// pointer order: see include/climsim_amd.h (csa_phys_wrapped_create)
extern "C" int csa_phys_wrapped_create(int nh, int ng, int flags, const float *const *w, int max_batch, csa_phys **out)
{
    if (!w || !out || max_batch <= 0 || (flags & ~(CSA_PHYS_LIQ_FRAC_HEAD | CSA_PHYS_STOCHASTIC | CSA_PHYS_RAD_UPDATED_QV | CSA_PHYS_SW_HEAD | CSA_PHYS_SW_GAS))) {
        csa_set_error_msg("csa_phys_wrapped_create: bad argument");
        return CSA_ERR_ARG;
    }
    if ((nh != 128 && nh != 112 && nh != 96) || !(ng == 12 || ng == 14 || ng == 16)) {
        csa_set_error_msg("csa_phys_wrapped_create: built for the shipped exports (GRU 128 / 112 / 96; 12, 14 or 16 regions = g-points)");
        return CSA_ERR_UNSUPPORTED;
    }
    PhysHostW v{};
    const float *const *p = w;
    v.hyam = *p++; v.hybm = *p++; v.hyai = *p++; v.hybi = *p++; v.ysl = *p++; v.yss = *p++; v.xds = *p++; v.xms = *p++;
    v.init_w = *p++; v.init_b = *p++; v.s1_w = *p++; v.s1_b = *p++;
    v.r1_ih = *p++; v.r1_hh = *p++; v.r1_bih = *p++; v.r1_bhh = *p++; v.r2_ih = *p++; v.r2_hh = *p++; v.r2_bih = *p++; v.r2_bhh = *p++;
    v.lat_w = *p++; v.lat_b = *p++; v.out_w = *p++; v.out_b = *p++; v.rel_w = *p++; v.rel_b = *p++;
    v.heads = p; p += 2 * PH_NHEAD;
    v.ys_rad = *p++; v.solar_w = *p++; v.g_xmin = *p++; v.g_xmax = *p++; v.g_ymean = *p++; v.g_ystd = *p++;
    v.g_w1 = *p++; v.g_b1 = *p++; v.g_w2 = *p++; v.g_b2 = *p++; v.g_w3 = *p++; v.g_b3 = *p++;
    v.r1_w = *p++; v.r1_b = *p++; v.r2_w = *p++; v.r2_b = *p++;
    v.nx21 = 1;
    if (flags & CSA_PHYS_SW_HEAD) { v.sw1_w = *p++; v.sw1_b = *p++; v.sw2_w = *p++; v.sw2_b = *p++; v.lbd_qn = *p++; }
    else if (flags & CSA_PHYS_SW_GAS) { v.swg = *p++; v.cldtab = *p++; }
    else { v.swx = *p++; v.cldtab = *p++; }
    v.misc = *p++;
    const float *xmean_lev = *p++, *xdiv_lev = *p++, *lqc = *p++, *lqi = *p++;
    if (flags & CSA_PHYS_LIQ_FRAC_HEAD) { v.liq_w = *p++; v.liq_b = *p++; }
    if (flags & CSA_PHYS_STOCHASTIC) { v.s3_ih = *p++; v.s3_zh = *p++; v.s3_enc = *p++; }
    for (const float *const *q = w; q != p; ++q)
        if (!*q) { csa_set_error_msg("csa_phys_wrapped_create: null weight pointer"); return CSA_ERR_ARG; }
    v.rad_qv_upd = (flags & CSA_PHYS_RAD_UPDATED_QV) ? 1 : 0;
    // the model behind the wrapper: 21 level inputs (18 + pressure feed mlp_initial), 19 surface inputs (0:6 and 11:19 feed mlp_surface1),
    // 16 regions (the caller zero-pads 12 / 14), 15 + 1 memory channels, later-export LW scheme
    int rc = phys_build(21, 18, 19, 14, 6, nh, 10, 16, 15, 1, 0, 1, v, max_batch, out);
    if (rc) return rc;
    csa_phys *h = *out;
    h->ng = ng;
    auto up = [&](const float *src, size_t n) {
        void *q = nullptr;
        if (hipMalloc(&q, sizeof(float) * n) != hipSuccess) { rc = CSA_ERR_NOMEM; return (float *)nullptr; }
        h->owned.push_back(q);
        if (src && hipMemcpy(q, src, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
        return (float *)q;
    };
    h->wr_xmean = up(xmean_lev, 60 * 21); h->wr_xdiv = up(xdiv_lev, 60 * 21); h->wr_lqc = up(lqc, 60); h->wr_lqi = up(lqi, 60);
    const size_t M = (size_t)PH_L * max_batch;
    h->XM = up(nullptr, M * 21); h->XD = up(nullptr, M * 21); h->XS = up(nullptr, (size_t)max_batch * 19);
    h->O5 = up(nullptr, M * 5); h->OS = up(nullptr, (size_t)max_batch * 8);
    if (rc) { csa_phys_destroy(h); *out = nullptr; }
    return rc;
}

int launch_phys_prep(const PhysDev &d, int B, const float *x_main, const float *x_sfc, const float *mem, float *X1, float *hx, hipStream_t s)
{
    hipLaunchKernelGGL(phys_prep_kernel, dim3(B, B <= 1024 ? 4 : 1), dim3(128), 0, s, d, B, x_main, x_sfc, mem, X1, hx);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

int launch_phys_decode_hidden(const PhysDev &d, int B, const float *HD, const float *Hlast, const float *x_sfc, const float *mem,
                              const float *x_denorm, int nxd, float *out_lev, float *out_sfc, float *mem_out, hipStream_t s)
{
    hipLaunchKernelGGL((phys_decode_kernel<16, 512, false>), dim3(B), dim3(512), 0, s, d, B, HD, Hlast, x_sfc, mem, x_denorm, nxd,
                       out_lev, out_sfc, mem_out, PhysRadOut{});
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

extern "C" int csa_phys_destroy(csa_phys *h)
{
    if (!h) return CSA_ERR_ARG;
    if (h->tr) phys_train_free(h->tr);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->rnn3) csa_stoch_destroy(h->rnn3);
    for (void *p : h->owned) (void)hipFree(p);
    delete h;
    return CSA_OK;
}

__global__ __launch_bounds__(256) void phys_mul_kernel(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, f32x4 *__restrict__ o, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) o[i] = a[i] * b[i];
}

// From CSA_PHYS_HALVES_MIN columns (default 640, the LSTM path's threshold: the deployed export measured 0.428 -> 0.380 ms at 640 columns,
// 0.776 -> 0.674 ms at 1,280, 1.391 -> 1.264 ms at 2,700; 0.297 -> 0.340 ms at 512) a physRNN call runs as two column halves on two
// streams, as the LSTM path does (api.hip::run_forward_halves): the decoder / optics / solver launches are chains of dependent work per
// workgroup, and two such chains overlap.  The second half works in the upper part of every work array (the handle's pointers are
// shifted for the duration of ITS launches: kernels take them by value) and addresses the caller's level-major tensors (rnn_mem,
// mem_out, mask_u) with the call's row stride (PhysDev::mem_B / mem_off).  Not with the stochastic third RNN (one layer handle with
// buffers of its own).  part(columns, first column, stream) launches one part.
static bool phys_halves_eligible(const csa_phys *h, int B)
{
    static const int halves_min = getenv("CSA_PHYS_HALVES_MIN") ? atoi(getenv("CSA_PHYS_HALVES_MIN")) : 640;
    return !h->rnn3 && halves_min > 0 && B >= halves_min;
}
template <typename F> static int phys_halves(csa_phys *h, int B, hipStream_t s, F part)
{
    if (!h->side) {
        CSA_HIP_CHECK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        CSA_HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        CSA_HIP_CHECK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    const int B1 = (B + 1) / 2, B2 = B - B1;
    CSA_HIP_CHECK(hipEventRecord(h->ev_fork, s));
    CSA_HIP_CHECK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    h->d.mem_B = B; h->d.mem_off = 0;
    int rc = part(B1, 0, s);
    if (rc == CSA_OK) {
        // per-column floats of every work array a call touches; the second half lives B1 columns further up
        const PhysDev &d = h->d;
        struct Shift { float **p; size_t per; } sh[] = {
            {&h->X1, (size_t)d.Lr * (d.nh + 16)}, {&h->P, (size_t)d.Lr * 4 * d.nh}, {&h->H1, (size_t)d.Lr * d.nh}, {&h->H2, (size_t)d.Lr * d.nh},
            {&h->hx, (size_t)d.nh}, {&h->HD, (size_t)d.Lr * d.hdw}, {&h->XG, (size_t)PH_L * PH_XG_K}, {&h->XR, (size_t)PH_L * PH_XR_K},
            {&h->RS, (size_t)PH_L * 2}, {&h->CL, (size_t)d.Lc * PH_NG}, {&h->TP, (size_t)PH_L * 32}, {&h->S2, (size_t)PH_L * 48},
            {&h->CS, (size_t)d.Lc * 48}, {&h->XM, (size_t)PH_L * 21}, {&h->XS, (size_t)19}, {&h->XD, (size_t)PH_L * 21}, {&h->O5, (size_t)PH_L * 5},
            {&h->OS, (size_t)8}};
        for (Shift &x : sh) if (*x.p) *x.p += x.per * B1;
        h->d.mem_off = B1;
        rc = part(B2, B1, h->side);
        for (Shift &x : sh) if (*x.p) *x.p -= x.per * B1;
    }
    h->d.mem_B = 0; h->d.mem_off = 0;
    if (rc) return rc;
    CSA_HIP_CHECK(hipEventRecord(h->ev_join, h->side));
    CSA_HIP_CHECK(hipStreamWaitEvent(s, h->ev_join, 0));
    return CSA_OK;
}

// x_main (B,60,nx) normalised, x_sfc (B,naux) normalised, rnn_mem (B,50,16), x_denorm (B,60,nxd) raw (T, ., qliq, qice, ..., qv last),
// hx2 (B,nh): the N(0,1) draw the reference makes for rnn2's initial state; add_stochastic_layer graphs also draw hx1 (B,nh), rnn3's
// initial state, and eps3 (Lr,B,nh), the layer's noise.  -> out_lev (B,60,5), out_sfc (B,8), mem_out (B,50,16)
static int phys_forward_part(csa_phys *h, int B, const float *x_main, const float *x_sfc, const float *rnn_mem,
                             const float *x_denorm, int nxd, const float *hx2, const float *hx1, const float *eps3, const float *srnn,
                             float *out_lev, float *out_sfc, float *mem_out, void *stream, const float *mask_u = nullptr)
{
    if (!h || !x_main || !x_sfc || !rnn_mem || !x_denorm || !hx2 || !out_lev || !out_sfc || !mem_out || B <= 0 || B > h->max_batch || nxd < 5 ||
        (h->d.rad && nxd < 16)) {
        csa_set_error_msg("csa_phys_forward: bad argument");
        return CSA_ERR_ARG;
    }
    if (h->rnn3 && !srnn && (!hx1 || !eps3)) {
        csa_set_error_msg("csa_phys_forward: this graph has the stochastic third RNN: pass its N(0,1) draws (csa_phys_forward_noise)");
        return CSA_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const PhysDev &d = h->d;
    const int nh = d.nh, L = d.Lr, M = L * B;
    int rc;
    // level slices per column: 1 / 2 / 3 / 4 / 5 slices measured 263 / 255 / 252 / 251 / 251 us per 384-column call (radiation graph)
    const int lsplit = B <= 1024 ? 4 : 1;
    hipLaunchKernelGGL(phys_prep_kernel, dim3(B, lsplit), dim3(128), 0, s, d, B, x_main, x_sfc, rnn_mem, h->X1, h->hx);
    CSA_HIP_CHECK(hipGetLastError());
    // one column per workgroup up to 256 columns, two from there, four on the matrix pipe from 544 (the LSTM path's thresholds)
    const bool four = gru4m_selected(nh, B);
    auto rec = [&](const float *whh, const float *whg, const float *whm, const float *bhn, const float *h0, float *Hout, int reverse) {
        return B <= 256 ? launch_rec1_gru(nh, whh, bhn, h->P, h0, Hout, B, L, reverse, s)
               : four   ? launch_rec4m_gru(nh, whm, bhn, h->P, h0, Hout, B, L, reverse, s)
                        : launch_rec2_gru(nh, whg, bhn, h->P, h0, Hout, B, L, reverse, s);
    };
    if ((rc = launch_proj_gemm(h->X1, h->wih1, h->bias1, h->P, M, 3 * nh, nh + 16, s, 0))) return rc;
    if ((rc = rec(h->whh1p, h->whh1g, h->whh1m, h->bhn1, h->hx, h->H1, 1))) return rc;
    if ((rc = launch_proj_gemm(h->H1, h->wih2, h->bias2, h->P, M, 3 * nh, nh, s, 0))) return rc;
    if ((rc = rec(h->whh2p, h->whh2g, h->whh2m, h->bhn2, hx2, h->H2, 0))) return rc;
    const float *Hhead = h->H2, *Hlast = h->H2;      // the sequence the heads read; the sequence whose last state feeds the release head
    if (h->rnn3) {                                    // rnn2's output times the stochastic layer's output; last state: the layer's own
        if (srnn) CSA_HIP_CHECK(hipMemcpyAsync(h->H3, srnn, sizeof(float) * (size_t)M * nh, hipMemcpyDeviceToDevice, s));
        else if ((rc = csa_stoch_gru5_forward(h->rnn3, L, B, h->H2, hx1, eps3, h->H3, stream))) return rc;
        const size_t n4 = (size_t)M * nh / 4;
        hipLaunchKernelGGL(phys_mul_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const f32x4 *)h->H2, (const f32x4 *)h->H3,
                           (f32x4 *)h->H2p, n4);
        CSA_HIP_CHECK(hipGetLastError());
        Hhead = h->H2p; Hlast = h->rnn3_last_mul ? h->H2p : h->H3;
    }
    if ((rc = launch_proj_gemm(Hhead, h->whead, h->bhead, h->HD, M, d.hdw, nh, s, 0))) return rc;
    if (d.rad) {
        PhysRadOut ro{x_main, h->XG, h->XR, h->RS, h->CL, h->CS};
        if (d.ncol == 4)
            hipLaunchKernelGGL((phys_decode_kernel<4, 256, true>), dim3(B), dim3(256), 0, s, d, B, h->HD, Hlast, x_sfc, rnn_mem, x_denorm, nxd,
                               out_lev, out_sfc, mem_out, ro);
        else
            hipLaunchKernelGGL((phys_decode_kernel<16, 512, true>), dim3(B), dim3(512), 0, s, d, B, h->HD, Hlast, x_sfc, rnn_mem, x_denorm, nxd,
                               out_lev, out_sfc, mem_out, ro);
        CSA_HIP_CHECK(hipGetLastError());
        return launch_phys_radiation(h, B, x_sfc, out_lev, out_sfc, s, mask_u);
    }
    hipLaunchKernelGGL((phys_decode_kernel<16, 512, false>), dim3(B), dim3(512), 0, s, d, B, h->HD, Hlast, x_sfc, rnn_mem, x_denorm, nxd,
                       out_lev, out_sfc, mem_out, PhysRadOut{});
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// the unfrozen graphs' entry points: one part, or two column halves (phys_halves)
static int phys_forward_impl(csa_phys *h, int B, const float *x_main, const float *x_sfc, const float *rnn_mem,
                             const float *x_denorm, int nxd, const float *hx2, const float *hx1, const float *eps3, const float *srnn,
                             float *out_lev, float *out_sfc, float *mem_out, void *stream)
{
    if (!h || B <= 0 || !phys_halves_eligible(h, B) || !x_main || !x_sfc || !rnn_mem || !x_denorm || !hx2 || !out_lev || !out_sfc || !mem_out)
        return phys_forward_part(h, B, x_main, x_sfc, rnn_mem, x_denorm, nxd, hx2, hx1, eps3, srnn, out_lev, out_sfc, mem_out, stream);
    const PhysDev &d = h->d;
    const int nx = d.nx, naux = d.naux, nh = d.nh, lm = d.memlm, Lc = d.Lc;
    return phys_halves(h, B, (hipStream_t)stream, [&](int Bp, int off, hipStream_t sp) {
        // batch-first tensors move by pointer; level-major rnn_mem / mem_out keep their base and take the call's row stride
        const size_t moff = lm ? 0 : (size_t)off * Lc * 16;
        return phys_forward_part(h, Bp, x_main + (size_t)off * PH_L * nx, x_sfc + (size_t)off * naux, rnn_mem + moff,
                                 x_denorm + (size_t)off * PH_L * nxd, nxd, hx2 + (size_t)off * nh, hx1, eps3, srnn,
                                 out_lev + (size_t)off * PH_L * 5, out_sfc + (size_t)off * 8, mem_out + moff, sp);
    });
}

extern "C" int csa_phys_forward_noise(csa_phys *h, int B, const float *x_main, const float *x_sfc, const float *rnn_mem,
                                      const float *x_denorm, int nxd, const float *hx2, const float *hx1, const float *eps3,
                                      float *out_lev, float *out_sfc, float *mem_out, void *stream)
{
    return phys_forward_impl(h, B, x_main, x_sfc, rnn_mem, x_denorm, nxd, hx2, hx1, eps3, nullptr, out_lev, out_sfc, mem_out, stream);
}

extern "C" int csa_phys_forward(csa_phys *h, int B, const float *x_main, const float *x_sfc, const float *rnn_mem,
                                const float *x_denorm, int nxd, const float *hx2, float *out_lev, float *out_sfc, float *mem_out,
                                void *stream)
{
    return phys_forward_impl(h, B, x_main, x_sfc, rnn_mem, x_denorm, nxd, hx2, nullptr, nullptr, nullptr, out_lev, out_sfc, mem_out, stream);
}

// Test hook (teacher forcing): the forward pass with rnn3's output (Lr, B, nh) supplied instead of computed -- the stages after a
// chaotic stochastic layer checked on their own (tests/test_physrnn_rad.py).  The layer itself: csa_phys_debug_rnn3.
extern "C" int csa_phys_debug_forward_srnn(csa_phys *h, int B, const float *x_main, const float *x_sfc, const float *rnn_mem,
                                           const float *x_denorm, int nxd, const float *hx2, const float *srnn,
                                           float *out_lev, float *out_sfc, float *mem_out, void *stream)
{
    if (!h || !h->rnn3 || !srnn) { csa_set_error_msg("csa_phys_debug_forward_srnn: needs a handle with the stochastic layer and its output"); return CSA_ERR_ARG; }
    return phys_forward_impl(h, B, x_main, x_sfc, rnn_mem, x_denorm, nxd, hx2, nullptr, nullptr, srnn, out_lev, out_sfc, mem_out, stream);
}

// Test hook: the handle's own rnn3 on caller-supplied input -- x (T, B, nh), h0 (B, nh), eps (T, B, nh) -> out (T, B, nh); T * B within
// 50 * max_batch.  With T = 1 and B = all (level, column) pairs this is every step of the layer from the reference's previous state.
extern "C" int csa_phys_debug_rnn3(csa_phys *h, int T, int B, const float *x, const float *h0, const float *eps, float *out, void *stream)
{
    if (!h || !h->rnn3) { csa_set_error_msg("csa_phys_debug_rnn3: this graph has no stochastic layer"); return CSA_ERR_ARG; }
    return csa_stoch_gru5_forward(h->rnn3, T, B, x, h0, eps, out, stream);
}

// physical_RNN_autoreg.postprocessing (the artefacts' exported method; rnn/models/models.py:273-339 with mp_mode 1): tendencies
// de-normalised, the cloud-water tendency split into liquid and ice by the temperature ramp at the UPDATED temperature.
// out (B,60,5) [dT, dqv, dqn, du, dv] normalised -> out6 (B,60,6) [dT, dqv, dqliq, dqice, du, dv] physical; out_sfc / yscale_sca.
__global__ __launch_bounds__(256) void phys_post_kernel(PhysDev d, int B, const float *__restrict__ out, const float *__restrict__ out_sfc,
                                                        const float *__restrict__ x_denorm, int nxd, float *__restrict__ out6,
                                                        float *__restrict__ out_sfc_d, int scrub_nan = 0)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B * PH_L) {
        const int L = i % PH_L;
        const float *o = out + (size_t)i * 5, *ys = d.yscale_lev + L * 5, *xd = x_denorm + (size_t)i * nxd;
        float v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = o[k] / ys[k];
        const float T_new = xd[0] + v[0] * 1200.0f;
        const float lf = fminf(fmaxf((T_new - 253.16f) * 0.05f, 0.0f), 1.0f);
        const float qn_new = (xd[2] + xd[3]) + v[2] * 1200.0f;
        float *r = out6 + (size_t)i * 6;
        r[0] = v[0]; r[1] = v[1];
        r[2] = (lf * qn_new - xd[2]) * 0.0008333333333333334f;
        r[3] = ((1.0f - lf) * qn_new - xd[3]) * 0.0008333333333333334f;
        r[4] = v[3]; r[5] = v[4];
        if (scrub_nan) {      // the wrapper's last line (rnn/utils.py:293): NaN -> 0 on the level outputs
#pragma unroll
            for (int k = 0; k < 6; ++k) r[k] = r[k] != r[k] ? 0.0f : r[k];
        }
    }
    if (i < B * 8) out_sfc_d[i] = out_sfc[i] / d.yscale_sca[i & 7];
}

extern "C" int csa_phys_postprocess(csa_phys *h, int B, const float *out, const float *out_sfc, const float *x_denorm, int nxd,
                                    float *out6, float *out_sfc_denorm, void *stream)
{
    if (!h || !out || !out_sfc || !x_denorm || !out6 || !out_sfc_denorm || B <= 0 || nxd < 4) {
        csa_set_error_msg("csa_phys_postprocess: bad argument");
        return CSA_ERR_ARG;
    }
    hipLaunchKernelGGL(phys_post_kernel, dim3((B * PH_L + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->d, B, out, out_sfc, x_denorm, nxd,
                       out6, out_sfc_denorm);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// The frozen export's call: raw x_main0 (B,60,20), x_sfc0 (B,19), rnn1_mem (50,B,16) -> out_lev (B,60,6) physical, out_sfc (B,8) physical,
// rnn1_mem (50,B,16).  The draws the export makes inside forward are arguments: hx2 (B,nh), with CSA_PHYS_STOCHASTIC hx1 (B,nh) and eps3
// (50,B,nh), and mask_u (60,B,ng): the uniform field of the SW humidity coin (< 0.5: the largest region's humidity).  srnn (test hook,
// nullable): the third RNN's output supplied.
extern "C" int csa_phys_wrapped_forward(csa_phys *h, int B, const float *x_main0, const float *x_sfc0, const float *rnn1_mem, const float *hx2,
                                        const float *hx1, const float *eps3, const float *mask_u, const float *srnn, float *out_lev,
                                        float *out_sfc, float *mem_out, void *stream)
{
    const bool coin = h && h->d.swg && !h->d.sw_e3sm;      // only the 7-32-32-ng SW gas-optics sub-generations flip the humidity coin
    if (!h || !h->XM || !x_main0 || !x_sfc0 || !rnn1_mem || !hx2 || (coin && !mask_u) || !out_lev || !out_sfc || !mem_out || B <= 0 || B > h->max_batch) {
        csa_set_error_msg("csa_phys_wrapped_forward: bad argument (or not a csa_phys_wrapped_create handle)");
        return CSA_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    auto part = [&](int Bp, int off, hipStream_t sp) -> int {      // columns [off, off + Bp) of the call on stream sp
        const PhysDev &d = h->d;
        hipLaunchKernelGGL(phys_wrap_pre_kernel, dim3((Bp * PH_L + 255) / 256), dim3(256), 0, sp, d, Bp, x_main0 + (size_t)off * PH_L * 20,
                           x_sfc0 + (size_t)off * 19, h->wr_xmean, h->wr_xdiv, h->wr_lqc, h->wr_lqi, h->XM, h->XS, h->XD);
        CSA_HIP_CHECK(hipGetLastError());
        int rc = phys_forward_part(h, Bp, h->XM, h->XS, rnn1_mem, h->XD, 21, hx2 + (size_t)off * d.nh, hx1, eps3, srnn, h->O5, h->OS, mem_out, sp, mask_u);
        if (rc) return rc;
        hipLaunchKernelGGL(phys_post_kernel, dim3((Bp * PH_L + 255) / 256), dim3(256), 0, sp, d, Bp, h->O5, h->OS, h->XD, 21,
                           out_lev + (size_t)off * PH_L * 6, out_sfc + (size_t)off * 8, 1);
        CSA_HIP_CHECK(hipGetLastError());
        return CSA_OK;
    };
    return phys_halves_eligible(h, B) ? phys_halves(h, B, s, part) : part(B, 0, s);
}

// taps for tests: level-major (Lr, B, nh) outputs of rnn1 (level order) and rnn2 of the last call
extern "C" int csa_phys_tap(csa_phys *h, int which, int B, float *dst, void *stream)
{
    if (h && dst && B > 0 && B <= h->max_batch && which >= 3 && which <= 8 && h->d.rad) {
        // radiation work arrays of the last call (level-major rows): 3 CS (50 B, 48), 4 CL (50 B, 16), 5 XR (60 B, 24), 6 S2 (60 B, 48),
        // 7 RS (60 B, 2), 8 TP (60 B, 32)
        const float *src[] = {h->CS, h->CL, h->XR, h->S2, h->RS, h->TP};
        const size_t n[] = {(size_t)h->d.Lc * B * 48, (size_t)h->d.Lc * B * PH_NG, (size_t)PH_L * B * PH_XR_K, (size_t)PH_L * B * 48,
                            (size_t)PH_L * B * 2, (size_t)PH_L * B * 32};
        if (!src[which - 3]) return CSA_ERR_ARG;
        CSA_HIP_CHECK(hipMemcpyAsync(dst, src[which - 3], sizeof(float) * n[which - 3], hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return CSA_OK;
    }
    if (!h || !dst || B <= 0 || B > h->max_batch || which < 1 || which > 2) return CSA_ERR_ARG;
    CSA_HIP_CHECK(hipMemcpyAsync(dst, which == 1 ? h->H1 : h->H2, sizeof(float) * (size_t)h->d.Lr * B * h->d.nh, hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
    return CSA_OK;
}
