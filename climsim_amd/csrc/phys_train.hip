// phys_train.hip -- backward of the physRNN "Hidden" path (SURVEY section 8 f#1: training of rnn/models/models_phys.py::physical_RNN_autoreg,
// instantiated as the trainable model at rnn/train_rnn_rollout_torchscript_hydra.py:553-554; the loop feeds it like RNN_autoreg,
// rnn/utils.py:1070-1137).  Geometry: the non-radiative graph of the shipped physRNN-Hidden_* artefacts (use_physrad = false: two Linear
// radiation heads; GRU 128/128 over the 60 levels, 16 regions, 15 + 1 memory channels).
//
//   phys_decode_bwd_kernel   backward of phys.hip::phys_decode_kernel<16, 512, false> (microphysics_decode, models_phys.py:414-748, and the
//                            small heads applied inside it): given d(out_lev), d(out_sfc), d(mem_out) it recomputes the column's forward
//                            state in LDS and walks it back -- area-weighted sums become broadcasts, the flux divergences couple a level
//                            to its neighbour below, the three clamps route the gradient to the branch that won (torch.maximum / relu
//                            sub-gradients), the two softmaxes (regions, levels) and the three mean-preserving rescalings have their
//                            closed forms -- to d(head-GEMM output), d(last hidden state), d(stored water) and per-column partial gradients
//                            of the heads that live inside the decoder (mlp_output, mlp_surface_output_rad, mlp_precip_release).
//   csa_phys_train_*         forward with saved activations, BPTT through the two GRUs (train_rec.hip), the projection / head GEMMs and
//                            their weight gradients (train_misc.hip), phys_prep backward; flat gradient in state_dict order.
// Pinned by torch autograd through the CPU restatement of the forward pass (itself pinned to the shipped artefacts): tests/test_physrnn_train.py.
#include "phys.h"
#include "train.h"
#include <string>
#include <algorithm>

#define PB_LC 50
#define PB_NC 16
#define PB_T 512
#define PB_CELLS (PB_LC * PB_NC)

// per-column partial gradients written by the decoder backward: [out_w 5*15][out_b 5][sfo_w 6*nh][sfo_b 6][rel_w nh][rel_b 1]
__host__ __device__ inline int pb_part_floats(int nh) { return 75 + 5 + 6 * nh + 6 + nh + 1; }

__global__ __launch_bounds__(PB_T) void phys_decode_bwd_kernel(PhysDev d, int B, const float *__restrict__ HD, const float *__restrict__ H2,
                                                              const float *__restrict__ x_sfc, const float *__restrict__ mem,
                                                              const float *__restrict__ x_denorm, int nxd,
                                                              const float *__restrict__ d_out, const float *__restrict__ d_out_sfc,
                                                              const float *__restrict__ d_mem_out,
                                                              float *__restrict__ dHD, float *__restrict__ d_last_h, float *__restrict__ d_pold,
                                                              float *__restrict__ part)
{
    constexpr int LC = PB_LC, NC = PB_NC, nm0 = 15;
    __shared__ float s_out[LC][5], s_pv[LC], s_sm[LC], s_pd[LC], s_red[16], s_scal[16], s_dscal[8];
    __shared__ float s_area[PB_CELLS], s_qv[PB_CELLS], s_qn[PB_CELLS], s_qi[PB_CELLS], s_fH[PB_CELLS], s_fqv[PB_CELLS], s_fqn[PB_CELLS],
        s_sed[PB_CELLS], s_T[PB_CELLS];
    __shared__ float g_fH[PB_CELLS], g_fqv[PB_CELLS], g_fqn[PB_CELLS], g_sed[PB_CELLS];     // k * d(divergence) of the level, per cell
    __shared__ float s_dpv[LC], s_dliq[LC], s_dS[LC], s_dso[LC][5], s_lat[LC][16];
    const int nh = d.nh, b = blockIdx.x, tid = threadIdx.x, ilev = d.ilev, HDW = d.hdw;
    const float CP = 1004.64f, G = 9.80665f, LV = 2510400.0f, LS = 2844000.0f, OOG = 0.1019716213f;
    const float sp = x_sfc[(size_t)b * d.naux] * d.xdiv_sca0 + d.xmean_sca0;
    const float P_old = mem[ph_mem_row(d, B, b, LC - 1) * (nm0 + 1) + nm0];
    const float *last_h = H2 + ((size_t)(d.Lr - 1) * B + b) * nh;
    float *pp = part + (size_t)b * pb_part_floats(nh);

    // ---------------- forward recomputation (phases A, B of phys_decode_kernel) ----------------
    for (int l = tid; l < LC; l += PB_T) {
        const float *hd = HD + ((size_t)(l + ilev) * B + b) * HDW + PH_NHEAD * NC;
#pragma unroll
        for (int k = 0; k < nm0; ++k) s_lat[l][k] = hd[k];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            float a = d.out_b[v];
#pragma unroll
            for (int k = 0; k < nm0; ++k) a = fmaf(hd[k], d.out_w[v * nm0 + k], a);
            s_out[l][v] = a;
        }
        const int L = l + ilev;
        s_pd[l] = sp * (d.hybi[L + 1] - d.hybi[L]) + (d.hyai[L + 1] - d.hyai[L]) * 100000.0f;
    }
    if (tid >= 64 && tid < 64 + 7 * 8) {
        const int o = (tid - 64) >> 3, part8 = tid & 7;
        const float *w = o < 6 ? d.sfo_w + o * nh : d.rel_w;
        float a = 0.0f;
        for (int k = 0; k < nh / 8; ++k) a = fmaf(last_h[part8 + 8 * k], w[part8 + 8 * k], a);
        a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
        if (part8 == 0) s_scal[o] = a + (o < 6 ? d.sfo_b[o] : d.rel_b[0]);
    }
    __syncthreads();
    if (tid < 64) {
        const float v = tid < LC ? s_out[tid][2] : -3.0e38f;
        float m = v;
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float e = tid < LC ? expf(v - m) : 0.0f;
        float s = e;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (tid < LC) { s_sm[tid] = e / s; s_pv[tid] = e / s * P_old; }
    }
    // per-cell forward values kept in registers for this thread's (up to) two cells
    float r_area[2], r_qv0[2], r_qn0[2], r_qi0[2], r_sv[2], r_sn[2], r_si[2], r_mqv[2], r_mqn[2], r_mqi[2], r_dTh[2], r_raw[2], r_flux1[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = it * PB_T + tid, l = e >> 4, c = e & 15;
        const bool ok = l < LC;
        const int L = (ok ? l : 0) + ilev;
        const float *hd = HD + ((size_t)L * B + b) * HDW + c;
        const float *xd = x_denorm + ((size_t)b * PH_L + L) * nxd;
        const float a_raw = hd[H_AREA * NC];
        const float am = ph_max<NC>(a_raw), ae = expf(a_raw - am), area = ae / ph_sum<NC>(ae);
        const float qv0 = ph_softplus(hd[H_QV * NC]), qn0 = ph_softplus(hd[H_QN * NC]), qi0 = ph_softplus(hd[H_QICE * NC]);
        const float mqv = ph_sum<NC>(qv0 * area), mqn = ph_sum<NC>(qn0 * area), mqi = ph_sum<NC>(qi0 * area);
        const float sv = mqv == 0.0f ? 1.0f : xd[nxd - 1] / mqv, sn = mqn == 0.0f ? 1.0f : (xd[2] + xd[3]) / mqn, si = mqi == 0.0f ? 1.0f : xd[3] / mqi;
        const float dTh = hd[H_T * NC];
        const float T_crm = xd[0] + (dTh - ph_sum<NC>(dTh * area));
        const float play = d.hyam[L] * 100000.0f + sp * d.hybm[L], play_up = d.hyam[L - 1] * 100000.0f + sp * d.hybm[L - 1];
        const float raw = hd[H_EDDY * NC] * (CP / G) * T_crm * (play - play_up);
        const float fH = l == LC - 1 ? -fmaxf(raw, 0.0f) : raw;
        const float flux1 = hd[H_FLUX * NC] * 300000.0f;
        r_area[it] = area; r_qv0[it] = qv0; r_qn0[it] = qn0; r_qi0[it] = qi0; r_sv[it] = sv; r_sn[it] = sn; r_si[it] = si;
        r_mqv[it] = mqv; r_mqn[it] = mqn; r_mqi[it] = mqi; r_dTh[it] = dTh; r_raw[it] = raw; r_flux1[it] = flux1;
        if (ok) {
            s_area[e] = area; s_qv[e] = qv0 * sv; s_qn[e] = qn0 * sn; s_qi[e] = qi0 * si; s_fH[e] = fH; s_T[e] = T_crm;
            s_fqv[e] = flux1 * qv0 * sv; s_fqn[e] = flux1 * qn0 * sn;
            s_sed[e] = fmaxf(hd[H_SED * NC], 0.0f) * G * (qi0 * si) * d.yscale_lev[L * 5 + 2];
        }
    }
    if (tid < LC) { s_dpv[tid] = 0.0f; s_dliq[tid] = 0.0f; s_dS[tid] = 0.0f; }
    __syncthreads();

    // forward values the water budget needs: precipitation production of every level (phase C's clamps, cell-parallel, summed over the
    // regions with the forward kernel's own shuffle tree) and the sedimentation reaching the surface
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = it * PB_T + tid, l = e >> 4, c = e & 15;
        const bool ok = l < LC;
        const int lc = ok ? l : 0, ec = ok ? e : c, L = lc + ilev;
        const float *hd = HD + ((size_t)L * B + b) * HDW + c;
        const float *ys = d.yscale_lev + L * 5;
        const float pd = s_pd[lc], qv = s_qv[ec], qn = s_qn[ec];
        const bool up = lc > 0, last = lc == LC - 1;
        const float fqv_dp = ((last ? 0.0f : s_fqv[ec]) - (up ? s_fqv[ec - NC] : 0.0f)) / pd * (-G);
        const float fqn_dp = ((last ? 0.0f : s_fqn[ec]) - (up ? s_fqn[ec - NC] : 0.0f)) / pd * (-G);
        const float sed_dp = (s_sed[ec] - (up ? s_sed[ec - NC] : 0.0f)) / pd * (-G);
        float evap = (fmaxf(hd[H_EVAP * NC], 0.0f) + 1e-6f) * s_pv[lc];
        float cond = hd[H_COND * NC];
        float aa = fmaxf(hd[H_AA * NC], 0.0f) * qn * ys[2];
        cond = fmaxf(cond, ((-(ys[2] * qn / 1200.0f) - fqn_dp) + aa) - sed_dp);
        evap = fmaxf(evap, (-(ys[1] * qv / 1200.0f) - fqv_dp) + cond);
        aa = fmaxf(aa, ((fqn_dp + cond) + sed_dp) - ys[2] * (-qn + 0.0006f) / 1200.0f);
        const float sprec = ph_sum<NC>(s_area[ec] * (aa - evap)), ssed = ph_sum<NC>(s_area[ec] * s_sed[ec]);
        if (ok && c == 0) { s_dS[lc] = pd * OOG * sprec; if (last) s_dliq[0] = ssed; }
    }
    __syncthreads();

    // ---------------- phase D': column water budget, surface heads ----------------
    if (tid < 64) {
        float w = tid < LC ? s_dS[tid] : 0.0f, ssed_last = tid == 0 ? s_dliq[0] : 0.0f;
        for (int o = 32; o > 0; o >>= 1) { w += __shfl_xor(w, o); ssed_last += __shfl_xor(ssed_last, o); }
        if (tid == 0) {
            const float *dos = d_out_sfc + (size_t)b * 8;
            float dst = 0.0f;
            for (int l = 0; l < LC; ++l) dst += d_mem_out[ph_mem_row(d, B, b, l) * (nm0 + 1) + nm0];
            const float water_pre = P_old + w, water_new = fmaxf(water_pre, 0.0f);
            const float rel = 1.0f / (1.0f + expf(-s_scal[6]));
            const float stored_pre = water_new * (1.0f - rel);
            const float Tsfc = x_denorm[((size_t)b * PH_L + (PH_L - 1)) * nxd];
            const float Pmax = d.yscale_sca[3] * 1000.0f * 5.58e-18f * expf(Tsfc * 0.077f);
            const float snowfrac = fminf(fmaxf((-Tsfc + 283.3f) / 14.6f, 0.0f), 1.0f);
            const float d_precc = dos[3] + dos[2] * snowfrac;
            float d_excess = d_precc / 1000.0f - dst;
            const float d_released = d_precc / 1000.0f, d_ssed = d_precc / 1000.0f;
            const float d_stored_pre = dst + (stored_pre - Pmax > 0.0f ? d_excess : 0.0f);
            const float d_water_new = d_stored_pre * (1.0f - rel) + d_released * rel;
            const float d_rel = (d_released - d_stored_pre) * water_new;
            const float d_PW = water_pre > 0.0f ? d_water_new : 0.0f;
            s_red[0] = d_PW;                              // d(dprec[l]) for every level, and the first part of d(P_old)
            s_red[1] = d_ssed;
            s_dscal[6] = d_rel * rel * (1.0f - rel);
            const int omap[6] = {0, 1, 4, 5, 6, 7};
            for (int k = 0; k < 6; ++k) s_dscal[k] = s_scal[k] > 0.0f ? dos[omap[k]] : 0.0f;
        }
    }
    __syncthreads();
    // d(last_h) and the surface heads' weight gradients
    for (int j = tid; j < nh; j += PB_T) {
        float a = s_dscal[6] * d.rel_w[j];
#pragma unroll
        for (int o = 0; o < 6; ++o) a = fmaf(s_dscal[o], d.sfo_w[o * nh + j], a);
        d_last_h[(size_t)b * nh + j] = a;
        const float h = last_h[j];
#pragma unroll
        for (int o = 0; o < 6; ++o) pp[80 + o * nh + j] = s_dscal[o] * h;
        pp[80 + 6 * nh + 6 + j] = s_dscal[6] * h;
    }
    if (tid < 6) pp[80 + 6 * nh + tid] = s_dscal[tid];
    if (tid == 6) pp[80 + 6 * nh + 6 + nh] = s_dscal[6];

    // ---------------- phase C': tendencies, clamps, divergences ----------------
    const float d_PW = s_red[0], d_ssed = s_red[1];
    float g_area_r[2], g_qv_r[2], g_qn_r[2], g_hd_evap[2], g_hd_cond[2], g_hd_aa[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = it * PB_T + tid, l = e >> 4, c = e & 15;
        const bool ok = l < LC;
        const int lc = ok ? l : 0, ec = ok ? e : c, L = lc + ilev;
        const float *hd = HD + ((size_t)L * B + b) * HDW + c;
        const float *ys = d.yscale_lev + L * 5;
        const float *dob = d_out + ((size_t)b * PH_L + L) * 5;
        const float pd = s_pd[lc], area = s_area[ec], qv = s_qv[ec], qn = s_qn[ec];
        const bool up = lc > 0, last = lc == LC - 1;
        const float kq = (-G) / pd, kt = (-G / CP) / pd;
        const float flux_t_dp = (s_fH[ec] - (up ? s_fH[ec - NC] : 0.0f)) * kt;
        const float fqv_dp = ((last ? 0.0f : s_fqv[ec]) - (up ? s_fqv[ec - NC] : 0.0f)) * kq;
        const float fqn_dp = ((last ? 0.0f : s_fqn[ec]) - (up ? s_fqn[ec - NC] : 0.0f)) * kq;
        const float sed_dp = (s_sed[ec] - (up ? s_sed[ec - NC] : 0.0f)) * kq;
        const float hE = hd[H_EVAP * NC], hA = hd[H_AA * NC];
        const float evap1 = (fmaxf(hE, 0.0f) + 1e-6f) * s_pv[lc], cond1 = hd[H_COND * NC], aa1 = fmaxf(hA, 0.0f) * qn * ys[2];
        const float X1 = ((-(ys[2] * qn / 1200.0f) - fqn_dp) + aa1) - sed_dp;
        const float cond2 = fmaxf(cond1, X1);
        const float X2 = (-(ys[1] * qv / 1200.0f) - fqv_dp) + cond2;
        const float evap2 = fmaxf(evap1, X2);
        const float X3 = ((fqn_dp + cond2) + sed_dp) - ys[2] * (-qn + 0.0006f) / 1200.0f;
        const float aa2 = fmaxf(aa1, X3);
        const float dqv = (fqv_dp - cond2) + evap2, dqn = ((fqn_dp + cond2) - aa2) + sed_dp;
        const float S = ph_sum<NC>(area * flux_t_dp);
        const float temp = x_denorm[((size_t)b * PH_L + L) * nxd] + (S / ys[0]) * 1200.0f;
        const float lraw = (temp - 253.16f) * 0.05f, liq = fminf(fmaxf(lraw, 0.0f), 1.0f);
        const float lat = liq * LV + (1.0f - liq) * LS;
        const float net = (lat * cond2 - evap2 * LV) * (1.0f / CP);
        const float dT_crm = flux_t_dp + net / ys[1] * ys[0];
        // upstream gradients of the level's area-weighted sums
        const float g_sT = dob[0], g_sqv = dob[1], g_sqn = dob[2], g_sprec = d_PW * pd * OOG;
        float g_area = g_sT * dT_crm + g_sqv * dqv + g_sqn * dqn + g_sprec * (aa2 - evap2) + (last ? d_ssed * s_sed[ec] : 0.0f);
        const float g_dT = g_sT * area, g_dqv = g_sqv * area, g_dqn = g_sqn * area;
        float g_ftd = g_dT;
        const float g_net = g_dT * ys[0] / ys[1];
        float g_cond2 = g_net * lat * (1.0f / CP) - g_dqv + g_dqn;
        float g_evap2 = -g_net * LV * (1.0f / CP) + g_dqv - g_sprec * area;
        float g_aa2 = -g_dqn + g_sprec * area;
        // liquid fraction is one value per level: reduce its gradient over the regions, then through the ramp to the level's S
        const float g_liq = ph_sum<NC>(g_net * (LV - LS) * cond2 * (1.0f / CP));
        const float g_temp = (lraw > 0.0f && lraw < 1.0f) ? g_liq * 0.05f : 0.0f;
        const float g_S = g_temp * 1200.0f / ys[0];
        g_ftd += g_S * area;
        g_area += g_S * flux_t_dp;
        float g_fqv_dp = g_dqv, g_fqn_dp = g_dqn, g_sed_dp = g_dqn, g_qv = 0.0f, g_qn = 0.0f, g_aa1 = 0.0f, g_evap1 = 0.0f, g_cond1 = 0.0f;
        if (aa1 >= X3) g_aa1 = g_aa2;
        else { g_fqn_dp += g_aa2; g_cond2 += g_aa2; g_sed_dp += g_aa2; g_qn += g_aa2 * ys[2] / 1200.0f; }
        if (evap1 >= X2) g_evap1 = g_evap2;
        else { g_qv -= g_evap2 * ys[1] / 1200.0f; g_fqv_dp -= g_evap2; g_cond2 += g_evap2; }
        if (cond1 >= X1) g_cond1 = g_cond2;
        else { g_qn -= g_cond2 * ys[2] / 1200.0f; g_fqn_dp -= g_cond2; g_aa1 += g_cond2; g_sed_dp -= g_cond2; }
        g_hd_aa[it] = hA > 0.0f ? g_aa1 * qn * ys[2] : 0.0f;
        g_qn += g_aa1 * fmaxf(hA, 0.0f) * ys[2];
        g_hd_cond[it] = g_cond1;
        g_hd_evap[it] = hE > 0.0f ? g_evap1 * s_pv[lc] : 0.0f;
        const float g_pv = ph_sum<NC>(g_evap1 * (fmaxf(hE, 0.0f) + 1e-6f));
        if (ok && c == 0) s_dpv[lc] = g_pv;
        if (ok) { g_fH[e] = g_ftd * kt; g_fqv[e] = g_fqv_dp * kq; g_fqn[e] = g_fqn_dp * kq; g_sed[e] = g_sed_dp * kq; }
        g_area_r[it] = g_area; g_qv_r[it] = g_qv; g_qn_r[it] = g_qn;
    }
    __syncthreads();

    // ---------------- phase B': fluxes, sub-grid state, the three rescalings, the region softmax ----------------
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = it * PB_T + tid, l = e >> 4, c = e & 15;
        const bool ok = l < LC;
        const int lc = ok ? l : 0, ec = ok ? e : c, L = lc + ilev;
        const float *hd = HD + ((size_t)L * B + b) * HDW + c;
        const bool last = lc == LC - 1;
        const float area = r_area[it];
        // flux at the bottom of layer l enters the divergence of l (+) and of l + 1 (-)
        const float below_H = last ? 0.0f : g_fH[ec + NC], below_qv = last ? 0.0f : g_fqv[ec + NC], below_qn = last ? 0.0f : g_fqn[ec + NC];
        const float below_sed = last ? 0.0f : g_sed[ec + NC];
        const float d_fH = g_fH[ec] - below_H;
        const float d_fqv = (last ? 0.0f : g_fqv[ec]) - below_qv, d_fqn = (last ? 0.0f : g_fqn[ec]) - below_qn;
        const float d_sed = (g_sed[ec] - below_sed) + (last ? d_ssed * area : 0.0f);
        const float ys2 = d.yscale_lev[L * 5 + 2];
        const float hS = hd[H_SED * NC], qi = r_qi0[it] * r_si[it];
        const float g_hd_sed = hS > 0.0f ? d_sed * G * qi * ys2 : 0.0f;
        const float g_qi = d_sed * fmaxf(hS, 0.0f) * G * ys2;
        const float qv = r_qv0[it] * r_sv[it], qn = r_qn0[it] * r_sn[it];
        const float g_flux1 = d_fqv * qv + d_fqn * qn;
        const float g_qv = g_qv_r[it] + d_fqv * r_flux1[it], g_qn = g_qn_r[it] + d_fqn * r_flux1[it];
        const float g_hd_flux = g_flux1 * 300000.0f;
        const float g_raw = last ? (r_raw[it] > 0.0f ? -d_fH : 0.0f) : d_fH;
        const float play = d.hyam[L] * 100000.0f + sp * d.hybm[L], play_up = d.hyam[L - 1] * 100000.0f + sp * d.hybm[L - 1];
        const float T_crm = s_T[ec], hEd = hd[H_EDDY * NC];
        const float g_hd_eddy = g_raw * (CP / G) * T_crm * (play - play_up);
        const float g_T = g_raw * hEd * (CP / G) * (play - play_up);
        const float sum_gT = ph_sum<NC>(g_T);
        const float g_hd_T = g_T - area * sum_gT;
        float g_area = g_area_r[it] - r_dTh[it] * sum_gT;
        // q = q0 * s, s = gcm / sum(q0 * area): d q0 = dq * s + d_m * area, d_m = -(sum dq q0) s / m  (s = 1 and d_m = 0 where m == 0)
        const float dsv = ph_sum<NC>(g_qv * r_qv0[it]), dsn = ph_sum<NC>(g_qn * r_qn0[it]), dsi = ph_sum<NC>(g_qi * r_qi0[it]);
        const float dmv = r_mqv[it] == 0.0f ? 0.0f : -dsv * r_sv[it] / r_mqv[it];
        const float dmn = r_mqn[it] == 0.0f ? 0.0f : -dsn * r_sn[it] / r_mqn[it];
        const float dmi = r_mqi[it] == 0.0f ? 0.0f : -dsi * r_si[it] / r_mqi[it];
        const float g_qv0 = g_qv * r_sv[it] + dmv * area, g_qn0 = g_qn * r_sn[it] + dmn * area, g_qi0 = g_qi * r_si[it] + dmi * area;
        g_area += dmv * r_qv0[it] + dmn * r_qn0[it] + dmi * r_qi0[it];
        const float g_araw = area * (g_area - ph_sum<NC>(area * g_area));
        auto dsoftplus = [](float x) { return x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x)); };
        if (ok) {
            float *o = dHD + ((size_t)L * B + b) * HDW + c;
            o[H_QV * NC] = g_qv0 * dsoftplus(hd[H_QV * NC]);
            o[H_QN * NC] = g_qn0 * dsoftplus(hd[H_QN * NC]);
            o[H_T * NC] = g_hd_T;
            o[H_AREA * NC] = g_araw;
            o[H_FLUX * NC] = g_hd_flux;
            o[H_EDDY * NC] = g_hd_eddy;
            o[H_QICE * NC] = g_qi0 * dsoftplus(hd[H_QICE * NC]);
            o[H_SED * NC] = g_hd_sed;
            o[H_EVAP * NC] = g_hd_evap[it];
            o[H_COND * NC] = g_hd_cond[it];
            o[H_AA * NC] = g_hd_aa[it];
        }
    }
    // ---------------- phase A': level softmax of the stored water, mlp_output, latent ----------------
    if (tid < 64) {
        const float gp = tid < LC ? s_dpv[tid] : 0.0f, sm = tid < LC ? s_sm[tid] : 0.0f;
        float dot = gp * sm;
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        if (tid < LC) {
            const int l = tid, L = l + ilev;
            const float *dob = d_out + ((size_t)b * PH_L + L) * 5;
            s_dso[l][0] = l >= 2 ? dob[0] : 0.0f;
            s_dso[l][1] = 0.0f;
            s_dso[l][2] = P_old * sm * (gp - dot);
            s_dso[l][3] = l >= 2 ? dob[3] : 0.0f;
            s_dso[l][4] = l >= 2 ? dob[4] : 0.0f;
        }
        if (tid == 0) d_pold[b] = d_PW + dot;       // d(P_old): through the water budget and through pv = softmax * P_old
    }
    __syncthreads();
    for (int i = tid; i < LC * 16; i += PB_T) {
        const int l = i >> 4, k = i & 15, L = l + ilev;
        float *o = dHD + ((size_t)L * B + b) * HDW;
        if (k < nm0) {
            float a = d_mem_out[ph_mem_row(d, B, b, l) * (nm0 + 1) + k];
#pragma unroll
            for (int v = 0; v < 5; ++v) a = fmaf(s_dso[l][v], d.out_w[v * nm0 + k], a);
            o[PH_NHEAD * NC + k] = a;
        } else {
            o[HDW - 1] = d_out[((size_t)b * PH_L + L) * 5];                       // the radiative heating head (every level)
        }
    }
    for (int i = tid; i < ilev * HDW; i += PB_T) {                                // levels above the CRM top: only that head
        const int L = i / HDW, k = i - L * HDW;
        dHD[((size_t)L * B + b) * HDW + k] = k == HDW - 1 ? d_out[((size_t)b * PH_L + L) * 5] : 0.0f;
    }
    // mlp_output partial gradients: [v][k] = sum_l d_s_out[l][v] lat[l][k]; bias = sum_l d_s_out[l][v]
    if (tid < 80) {
        const int v = tid < 75 ? tid / nm0 : tid - 75, k = tid < 75 ? tid - v * nm0 : -1;
        float a = 0.0f;
        for (int l = 0; l < LC; ++l) a = fmaf(s_dso[l][v], k >= 0 ? s_lat[l][k] : 1.0f, a);
        pp[tid] = a;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// phys_prep backward (mlp_initial through tanh, mlp_surface1 through tanh, gradient w.r.t. the incoming memory) as DATA for the
// split-M TN GEMMs that form the weight gradients: DPRE (M, nh) = dX1[:, :nh] * (1 - X1^2), XIN (M, 32) the layer's inputs (rows in
// rnn1's flipped level order, like X1), DHX (B, nh), XS (B, 32).  d_mem_in (B, 50, 16): channels 0..14 from dX1's memory columns,
// channel 15 (stored water) from the decoder backward at the lowest level.
__global__ __launch_bounds__(128) void phys_prep_bwd_kernel(PhysDev d, int B, const float *__restrict__ x_main, const float *__restrict__ x_sfc,
                                                            const float *__restrict__ X1, const float *__restrict__ hx,
                                                            const float *__restrict__ dX1, const float *__restrict__ dhx,
                                                            const float *__restrict__ d_pold, float *__restrict__ DPRE, float *__restrict__ XIN,
                                                            float *__restrict__ DHX, float *__restrict__ XS, float *__restrict__ d_mem_in)
{
    const int b = blockIdx.x, j = threadIdx.x, nh = d.nh, K1 = nh + 16;
    const float sp = x_sfc[(size_t)b * d.naux] * d.xdiv_sca0 + d.xmean_sca0;
    for (int l = 0; l < PH_L; ++l) {
        const size_t row = (size_t)(PH_L - 1 - l) * B + b;
        const float y = X1[row * K1 + j];
        DPRE[row * nh + j] = dX1[row * K1 + j] * (1.0f - y * y);
    }
    for (int i = j; i < PH_L * 32; i += 128) {
        const int l = i >> 5, k = i & 31;
        float v = 0.0f;
        if (k < d.nfeat) v = x_main[((size_t)b * PH_L + l) * d.nx + k];
        else if (k == d.nfeat) v = sqrtf(d.hyam[l] * 100000.0f + sp * d.hybm[l]) / 314.0f;
        XIN[((size_t)(PH_L - 1 - l) * B + b) * 32 + k] = v;
    }
    { const float y = hx[(size_t)b * nh + j]; DHX[(size_t)b * nh + j] = dhx[(size_t)b * nh + j] * (1.0f - y * y); }
    if (j < 32) XS[(size_t)b * 32 + j] = j < d.nx_sfc ? x_sfc[(size_t)b * d.naux + (j < d.sfc_cut ? j : j + d.sfc_skip)] : 0.0f;
    for (int i = j; i < d.Lc * 16; i += 128) {
        const int l = i >> 4, k = i & 15;
        d_mem_in[ph_mem_row(d, B, b, l) * 16 + k] = k < d.nm0 ? dX1[((size_t)(PH_L - 1 - (l + d.ilev)) * B + b) * K1 + nh + k]
                                                              : (l == d.Lc - 1 ? d_pold[b] : 0.0f);
    }
}

// dH[(L-1) rows][b][:] += d_last_h: the release / surface heads read rnn2's last state
__global__ void phys_add_last_kernel(float *__restrict__ dH_last, const float *__restrict__ dl, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dH_last[i] += dl[i];
}

struct PtInfo { std::string name; int off, rows, cols; };
struct PhysTrain {
    int nparam = 0, nsplit = 192, adam_step = 0;
    std::vector<PtInfo> info;
    PhysDev d;
    float *params = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    float *wih1, *bias1, *bhn1, *whh1g, *wih2, *bias2, *bhn2, *whh2g, *whead, *bhead;      // forward layouts (gate rows u*4 + [r, z, n, pad])
    float *whh1m, *whh2m;                                                                   // matrix-pipe GRU kernel (from 544 columns)
    float *wih1T, *wih2T, *whh1Tp, *whh2Tp, *wheadT;                                       // backward layouts
    struct Slot { float *X1, *H1, *hx, *HD, *GP1, *GP2, *Hs1, *Hs2; int B = 0; };      // what one pending forward keeps
    std::vector<Slot> slots;
    float *dHD, *dH2, *dH1, *dX1, *dlast, *dhx1, *dhx2, *dpold, *XIN, *XS, *DHX, *part, *part_b, *rtmp;
    // weight-gradient GEMMs: every one keeps its partials in a region of its own (arena) until ONE queued reduction at the end of the
    // backward call; they run on a side stream beside the dependent chain (BPTT kernels, dX GEMMs), which never reads them
    float *arena = nullptr;
    size_t arena_floats = 0;
    ReduceJobs rq;
    hipStream_t side = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float *samp, *ecoef, *sp;          // loss scratch (train_misc.hip::launch_loss), for nslots * max_batch samples
    int *m_whead, *m_bhead, *m_wih1, *m_whh1, *m_b1a, *m_b1b, *m_wih2, *m_whh2, *m_b2a, *m_b2b, *m_dec, *m_init, *m_initb, *m_s1, *m_s1b;
    std::vector<GatherEntry> gathers;
    GatherEntry *gtab = nullptr;
    int gmax = 0;
    std::vector<void *> owned;
};

void phys_train_free(PhysTrain *t)
{
    if (!t) return;
    for (void *p : t->owned) (void)hipFree(p);
    for (hipEvent_t e : t->ev) if (e) (void)hipEventDestroy(e);
    if (t->side) (void)hipStreamDestroy(t->side);
    delete t;
}

namespace {
template <typename T> T *pt_alloc(PhysTrain *t, size_t n, int &rc)
{
    void *p = nullptr;
    if (rc != CSA_OK) return nullptr;
    if (hipMalloc(&p, sizeof(T) * (n ? n : 1)) != hipSuccess) { rc = CSA_ERR_NOMEM; return nullptr; }
    t->owned.push_back(p);
    return (T *)p;
}
int *pt_map(PhysTrain *t, const std::vector<int> &m, int &rc)
{
    int *p = pt_alloc<int>(t, m.size(), rc);
    if (p && hipMemcpy(p, m.data(), sizeof(int) * m.size(), hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
    return p;
}
float *pt_gather(PhysTrain *t, const std::vector<int> &idx, const std::vector<int> *idx2, int &rc)
{
    float *dst = pt_alloc<float>(t, idx.size(), rc);
    int *i1 = pt_map(t, idx, rc), *i2 = idx2 ? pt_map(t, *idx2, rc) : nullptr;
    t->gathers.push_back({dst, i1, i2, (int)idx.size()});
    if ((int)idx.size() > t->gmax) t->gmax = (int)idx.size();
    return dst;
}
std::vector<int> pt_to_int(const std::vector<float> &f, int off)
{
    std::vector<int> v(f.size());
    for (size_t i = 0; i < f.size(); ++i) v[i] = off + (int)f[i];
    return v;
}
int pt_repack(PhysTrain *t, hipStream_t s) { return launch_gather_multi(t->gtab, (int)t->gathers.size(), t->gmax, t->params, s); }
}   // namespace

extern "C" int csa_phys_train_enable(csa_phys *h, int nslots)
{
    if (!h || nslots < 1 || nslots > 16) return CSA_ERR_ARG;
    if (h->tr) return (int)h->tr->slots.size() == nslots ? CSA_OK : CSA_ERR_ARG;
    if (h->d.rad || h->host_params.empty()) {
        csa_set_error_msg("csa_phys_train_enable: training is built for the non-radiative Hidden graph (csa_phys_create)");
        return CSA_ERR_UNSUPPORTED;
    }
    if (h->d.nx_sfc > 32) { csa_set_error_msg("csa_phys_train_enable: at most 32 surface inputs"); return CSA_ERR_UNSUPPORTED; }
    const PhysDev &d0 = h->d;
    const int nh = d0.nh, nf1 = d0.nfeat + 1, nxs = d0.nx_sfc, nm0 = d0.nm0, NC = d0.ncol, HDW = d0.hdw, Kin = nh + nm0, K1 = nh + 16;
    PhysTrain *t = new PhysTrain();
    int rc = CSA_OK, off = 0;
    auto add = [&](const std::string &name, int rows, int cols) { t->info.push_back({name, off, rows, cols}); off += rows * cols; };
    add("mlp_initial.weight", nh, nf1); add("mlp_initial.bias", nh, 1); add("mlp_surface1.weight", nh, nxs); add("mlp_surface1.bias", nh, 1);
    for (int r = 1; r <= 2; ++r) {
        const std::string p = "rnn" + std::to_string(r) + ".";
        add(p + "weight_ih_l0", 3 * nh, r == 1 ? Kin : nh); add(p + "weight_hh_l0", 3 * nh, nh);
        add(p + "bias_ih_l0", 3 * nh, 1); add(p + "bias_hh_l0", 3 * nh, 1);
    }
    add("mlp_latent.weight", nm0, nh); add("mlp_latent.bias", nm0, 1); add("mlp_output.weight", 5, nm0); add("mlp_output.bias", 5, 1);
    add("mlp_surface_output_rad.weight", 6, nh); add("mlp_surface_output_rad.bias", 6, 1);
    add("mlp_output_rad.weight", 1, nh); add("mlp_output_rad.bias", 1, 1);
    add("mlp_precip_release.weight", 1, nh); add("mlp_precip_release.bias", 1, 1);
    static const char *kHeads[PH_NHEAD] = {"mlp_qv_crm", "mlp_qn_crm", "mlp_t_crm", "mlp_subgrid_area_frac", "mlp_massflux", "mlp_eddy_diff",
                                           "mlp_qice_crm", "mlp_sed_qn_crm", "mlp_evap_prec_crm", "mlp_evap_cond_vapor_crm", "mlp_mp_aa_crm"};
    for (int k = 0; k < PH_NHEAD; ++k) { add(std::string(kHeads[k]) + ".weight", NC, nh); add(std::string(kHeads[k]) + ".bias", NC, 1); }
    t->nparam = off;
    if ((size_t)off != h->host_params.size()) { delete t; csa_set_error_msg("csa_phys_train_enable: parameter table mismatch"); return CSA_ERR_ARG; }
    auto O = [&](int i) { return t->info[i].off; };
    t->params = pt_alloc<float>(t, off, rc); t->adam_m = pt_alloc<float>(t, off, rc); t->adam_v = pt_alloc<float>(t, off, rc);
    if (rc == CSA_OK) {
        if (hipMemcpy(t->params, h->host_params.data(), sizeof(float) * off, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
        if (hipMemset(t->adam_m, 0, sizeof(float) * off) != hipSuccess || hipMemset(t->adam_v, 0, sizeof(float) * off) != hipSuccess) rc = CSA_ERR_HIP;
    }
    // ---- gathers: canonical flat -> kernel layouts; the same index arrays serve as scatter maps of the gradients ----
    t->d = d0;
    PhysDev &d = t->d;
    auto iota = [](int o, int n) { std::vector<int> v(n); for (int i = 0; i < n; ++i) v[i] = o + i; return v; };
    auto tr_idx = [](int o, int rows, int cols) {      // (rows, cols) row-major -> (cols, rows)
        std::vector<int> v((size_t)rows * cols);
        for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) v[(size_t)c * rows + r] = o + r * cols + c;
        return v;
    };
    d.init_wt = pt_gather(t, tr_idx(O(0), nh, nf1), nullptr, rc); d.init_b = pt_gather(t, iota(O(1), nh), nullptr, rc);
    d.s1_wt = pt_gather(t, tr_idx(O(2), nh, nxs), nullptr, rc); d.s1_b = pt_gather(t, iota(O(3), nh), nullptr, rc);
    d.out_w = pt_gather(t, iota(O(14), 5 * nm0), nullptr, rc); d.out_b = pt_gather(t, iota(O(15), 5), nullptr, rc);
    d.sfo_w = pt_gather(t, iota(O(16), 6 * nh), nullptr, rc); d.sfo_b = pt_gather(t, iota(O(17), 6), nullptr, rc);
    d.rel_w = pt_gather(t, iota(O(20), nh), nullptr, rc); d.rel_b = pt_gather(t, iota(O(21), 1), nullptr, rc);
    {   // mlp_initial / mlp_surface1 weight gradients arrive as (nh, 32) from the TN GEMM
        std::vector<int> mi((size_t)nh * 32, -1), ms((size_t)nh * 32, -1);
        for (int j = 0; j < nh; ++j) {
            for (int k = 0; k < nf1; ++k) mi[(size_t)j * 32 + k] = O(0) + j * nf1 + k;
            for (int k = 0; k < nxs; ++k) ms[(size_t)j * 32 + k] = O(2) + j * nxs + k;
        }
        t->m_init = pt_map(t, mi, rc); t->m_initb = pt_map(t, iota(O(1), nh), rc);
        t->m_s1 = pt_map(t, ms, rc); t->m_s1b = pt_map(t, iota(O(3), nh), rc);
    }
    // GRU layers: gate rows u*4 + [r, z, n, pad]; K of rnn1 padded from nh + 15 to nh + 16 (zero column).  After BPTT the saved-gate
    // buffer holds [dr~, dz~, dn~, g_hn] per unit: W_ih / b_ih gradients take columns 0, 1, 2; W_hh / b_hh take 0, 1 and 3 (as gate n).
    auto gru = [&](int Ksrc, int K, int o_wih, int o_whh, int o_bih, int o_bhh, float *&wih, float *&bias, float *&bhn, float *&whhg,
                   float *&whhTp, float *&wihT, int *&m_wih, int *&m_whh, int *&m_ba, int *&m_bb, float *&whhm) {
        std::vector<int> iw((size_t)4 * nh * K, -1), b1(4 * nh, -1), b2(4 * nh, -1), ibhn(nh), tt((size_t)K * 4 * nh, -1);
        std::vector<int> gw((size_t)4 * nh * nh, -1), gbb(4 * nh, -1);
        for (int u = 0; u < nh; ++u) {
            for (int g = 0; g < 3; ++g) {
                const int src = g * nh + u, dst = u * 4 + g;
                for (int k = 0; k < Ksrc; ++k) { iw[(size_t)dst * K + k] = o_wih + src * Ksrc + k; tt[(size_t)k * 4 * nh + dst] = o_wih + src * Ksrc + k; }
                b1[dst] = o_bih + src;
                if (g < 2) {
                    b2[dst] = o_bhh + src; gbb[dst] = o_bhh + src;
                    for (int k = 0; k < nh; ++k) gw[(size_t)dst * nh + k] = o_whh + src * nh + k;
                }
            }
            ibhn[u] = o_bhh + 2 * nh + u;
            gbb[u * 4 + 3] = o_bhh + 2 * nh + u;
            for (int k = 0; k < nh; ++k) gw[(size_t)(u * 4 + 3) * nh + k] = o_whh + (2 * nh + u) * nh + k;
        }
        wih = pt_gather(t, iw, nullptr, rc); bias = pt_gather(t, b1, &b2, rc); bhn = pt_gather(t, ibhn, nullptr, rc);
        wihT = pt_gather(t, tt, nullptr, rc);
        m_wih = pt_map(t, iw, rc); m_ba = pt_map(t, b1, rc); m_bb = pt_map(t, gbb, rc); m_whh = pt_map(t, gw, rc);
        std::vector<float> ih((size_t)3 * nh * nh), pk(rec_packed_floats(0, nh)), pkT(bwd_rec_packed_floats_gru(nh));
        for (size_t i = 0; i < ih.size(); ++i) ih[i] = (float)i;
        gru2_pack_weights(nh, ih.data(), pk.data());
        whhg = pt_gather(t, pt_to_int(pk, o_whh), nullptr, rc);
        bwd_rec_pack_weights_gru(nh, ih.data(), pkT.data());
        whhTp = pt_gather(t, pt_to_int(pkT, o_whh), nullptr, rc);
        {   // gru4m_pack_weights: zero fourth row of every block -> index -1
            std::vector<float> pm((size_t)4 * nh * nh);
            for (size_t i = 0; i < ih.size(); ++i) ih[i] = (float)(i + 1);
            gru4m_pack_weights(nh, ih.data(), pm.data());
            std::vector<int> im(pm.size());
            for (size_t i = 0; i < pm.size(); ++i) im[i] = pm[i] == 0.0f ? -1 : o_whh + (int)pm[i] - 1;
            whhm = pt_gather(t, im, nullptr, rc);
        }
    };
    gru(Kin, K1, O(4), O(5), O(6), O(7), t->wih1, t->bias1, t->bhn1, t->whh1g, t->whh1Tp, t->wih1T, t->m_wih1, t->m_whh1, t->m_b1a, t->m_b1b, t->whh1m);
    gru(nh, nh, O(8), O(9), O(10), O(11), t->wih2, t->bias2, t->bhn2, t->whh2g, t->whh2Tp, t->wih2T, t->m_wih2, t->m_whh2, t->m_b2a, t->m_b2b, t->whh2m);
    {   // head GEMM rows: 11 decoder heads x 16 regions, mlp_latent (15), mlp_output_rad (1)
        std::vector<int> mw((size_t)HDW * nh, -1), mb(HDW, -1), mt((size_t)nh * HDW, -1);
        for (int r = 0; r < HDW; ++r) {
            int ow, ob;
            if (r < PH_NHEAD * NC) { ow = O(22 + 2 * (r / NC)) + (r % NC) * nh; ob = O(23 + 2 * (r / NC)) + r % NC; }
            else if (r < PH_NHEAD * NC + nm0) { ow = O(12) + (r - PH_NHEAD * NC) * nh; ob = O(13) + r - PH_NHEAD * NC; }
            else { ow = O(18); ob = O(19); }
            mb[r] = ob;
            for (int j = 0; j < nh; ++j) { mw[(size_t)r * nh + j] = ow + j; mt[(size_t)j * HDW + r] = ow + j; }
        }
        t->whead = pt_gather(t, mw, nullptr, rc); t->bhead = pt_gather(t, mb, nullptr, rc); t->wheadT = pt_gather(t, mt, nullptr, rc);
        t->m_whead = pt_map(t, mw, rc); t->m_bhead = pt_map(t, mb, rc);
    }
    {   // decoder-backward partials: [out_w 75][out_b 5][sfo_w 6 nh][sfo_b 6][rel_w nh][rel_b 1]
        std::vector<int> m;
        auto app = [&](int o, int n) { for (int i = 0; i < n; ++i) m.push_back(o + i); };
        app(O(14), 5 * nm0); app(O(15), 5); app(O(16), 6 * nh); app(O(17), 6); app(O(20), nh); app(O(21), 1);
        t->m_dec = pt_map(t, m, rc);
    }
    t->gtab = pt_alloc<GatherEntry>(t, t->gathers.size(), rc);
    if (rc == CSA_OK && hipMemcpy(t->gtab, t->gathers.data(), sizeof(GatherEntry) * t->gathers.size(), hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
    // ---- saved activations and work arrays ----
    const size_t MB = (size_t)h->max_batch, M = (size_t)PH_L * MB;
    t->slots.resize(nslots);
    for (PhysTrain::Slot &S : t->slots) {
        S.X1 = pt_alloc<float>(t, M * K1, rc); S.H1 = pt_alloc<float>(t, M * nh, rc); S.hx = pt_alloc<float>(t, MB * nh, rc);
        S.HD = pt_alloc<float>(t, M * HDW, rc);
        S.GP1 = pt_alloc<float>(t, M * 4 * nh, rc); S.GP2 = pt_alloc<float>(t, M * 4 * nh, rc);
        S.Hs1 = pt_alloc<float>(t, (M + MB) * nh, rc); S.Hs2 = pt_alloc<float>(t, (M + MB) * nh, rc);
    }
    t->dHD = pt_alloc<float>(t, M * HDW, rc); t->dH2 = pt_alloc<float>(t, M * nh, rc); t->dH1 = pt_alloc<float>(t, M * nh, rc);
    t->dX1 = pt_alloc<float>(t, M * K1, rc); t->dlast = pt_alloc<float>(t, MB * nh, rc); t->dhx1 = pt_alloc<float>(t, MB * nh, rc);
    t->dhx2 = pt_alloc<float>(t, MB * nh, rc); t->dpold = pt_alloc<float>(t, MB, rc);
    t->XIN = pt_alloc<float>(t, M * 32, rc); t->XS = pt_alloc<float>(t, MB * 32, rc); t->DHX = pt_alloc<float>(t, MB * nh, rc);
    size_t pf = (size_t)t->nsplit * 4 * nh * K1;
    pf = std::max(pf, MB * (size_t)pb_part_floats(nh));
    t->part = pt_alloc<float>(t, pf, rc); t->part_b = pt_alloc<float>(t, (size_t)t->nsplit * 4 * nh, rc);
    t->rtmp = pt_alloc<float>(t, (size_t)32 * pb_part_floats(nh), rc);
    t->arena_floats = (size_t)t->nsplit * ((size_t)HDW * nh + HDW + 3 * (size_t)4 * nh * nh + (size_t)4 * nh * K1 + 2 * 4 * nh + 2 * ((size_t)nh * 32 + nh));
    t->arena = pt_alloc<float>(t, t->arena_floats, rc);
    t->samp = pt_alloc<float>(t, (size_t)nslots * MB * 9, rc); t->ecoef = pt_alloc<float>(t, MB, rc); t->sp = pt_alloc<float>(t, (size_t)nslots * MB, rc);
    if (rc == CSA_OK) rc = pt_repack(t, nullptr);
    if (rc == CSA_OK && hipDeviceSynchronize() != hipSuccess) rc = CSA_ERR_HIP;
    if (rc != CSA_OK) { phys_train_free(t); return rc; }
    h->tr = t;
    return CSA_OK;
}

extern "C" int csa_phys_train_num_params(csa_phys *h, int *n_tensors, int *n_floats)
{
    if (!h || !h->tr || !n_tensors || !n_floats) { csa_set_error_msg("csa_phys_train_num_params: call csa_phys_train_enable first"); return CSA_ERR_ARG; }
    *n_tensors = (int)h->tr->info.size(); *n_floats = h->tr->nparam;
    return CSA_OK;
}

extern "C" int csa_phys_train_param_info(csa_phys *h, int i, const char **name, int *offset, int *rows, int *cols)
{
    if (!h || !h->tr || i < 0 || i >= (int)h->tr->info.size()) return CSA_ERR_ARG;
    const PtInfo &p = h->tr->info[i];
    if (name) *name = p.name.c_str();
    if (offset) *offset = p.off;
    if (rows) *rows = p.rows;
    if (cols) *cols = p.cols;
    return CSA_OK;
}

// flat parameter vector (device pointers, state_dict order): read it out / replace it (re-packs the kernel layouts)
extern "C" int csa_phys_train_get_params(csa_phys *h, float *dst, void *stream)
{
    if (!h || !h->tr || !dst) return CSA_ERR_ARG;
    CSA_HIP_CHECK(hipMemcpyAsync(dst, h->tr->params, sizeof(float) * h->tr->nparam, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CSA_OK;
}
extern "C" int csa_phys_train_set_params(csa_phys *h, const float *src, void *stream)
{
    if (!h || !h->tr || !src) return CSA_ERR_ARG;
    CSA_HIP_CHECK(hipMemcpyAsync(h->tr->params, src, sizeof(float) * h->tr->nparam, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return pt_repack(h->tr, (hipStream_t)stream);
}

// training forward: as csa_phys_forward, through the training layouts, keeping what the backward pass needs (GRU gates and
// hidden sequences, X1, the head GEMM's output) in `slot`: one pending forward per slot (a TBPTT window of T_w steps uses T_w slots).
extern "C" int csa_phys_train_forward(csa_phys *h, int slot, int B, const float *x_main, const float *x_sfc, const float *rnn_mem,
                                      const float *x_denorm, int nxd, const float *hx2, float *out_lev, float *out_sfc, float *mem_out, void *stream)
{
    if (!h || !h->tr || slot < 0 || slot >= (int)h->tr->slots.size() || !x_main || !x_sfc || !rnn_mem || !x_denorm || !hx2 || !out_lev || !out_sfc || !mem_out || B <= 0 || B > h->max_batch || nxd < 5) {
        csa_set_error_msg("csa_phys_train_forward: bad argument (csa_phys_train_enable first)");
        return CSA_ERR_ARG;
    }
    PhysTrain *t = h->tr;
    hipStream_t s = (hipStream_t)stream;
    const PhysDev &d = t->d;
    const int nh = d.nh, L = PH_L, M = L * B;
    PhysTrain::Slot &S = t->slots[slot];
    int rc;
    S.B = 0;
    if ((rc = launch_phys_prep(d, B, x_main, x_sfc, rnn_mem, S.X1, S.hx, s))) return rc;
    if ((rc = launch_proj_gemm(S.X1, t->wih1, t->bias1, S.GP1, M, 4 * nh, nh + 16, s))) return rc;
    const bool four = gru4m_selected(nh, B);      // four columns per workgroup on the matrix pipe from 544 columns (as inference)
    if ((rc = four ? launch_rec4m_train_gru(nh, t->whh1m, t->bhn1, S.GP1, S.hx, S.H1, B, L, 1, S.Hs1, s)
                   : launch_rec_train_gru(nh, t->whh1g, t->bhn1, S.GP1, S.hx, S.H1, B, L, 1, S.Hs1, s))) return rc;
    if ((rc = launch_proj_gemm(S.H1, t->wih2, t->bias2, S.GP2, M, 4 * nh, nh, s))) return rc;
    float *H2 = S.Hs2 + (size_t)B * nh;
    if ((rc = four ? launch_rec4m_train_gru(nh, t->whh2m, t->bhn2, S.GP2, hx2, H2, B, L, 0, S.Hs2, s)
                   : launch_rec_train_gru(nh, t->whh2g, t->bhn2, S.GP2, hx2, H2, B, L, 0, S.Hs2, s))) return rc;
    if ((rc = launch_proj_gemm(H2, t->whead, t->bhead, S.HD, M, d.hdw, nh, s))) return rc;
    if ((rc = launch_phys_decode_hidden(d, B, S.HD, H2, x_sfc, rnn_mem, x_denorm, nxd, out_lev, out_sfc, mem_out, s))) return rc;
    S.B = B;
    return CSA_OK;
}

// backward of the pending forward: grads (nparam floats, state_dict order) += dLoss/dparams; d_mem_in (B, 50, 16) = dLoss/d(rnn_mem).
// The inputs are the forward's (they are read again, not copied).
extern "C" int csa_phys_train_backward(csa_phys *h, int slot, int B, const float *x_main, const float *x_sfc, const float *rnn_mem, const float *x_denorm,
                                       int nxd, const float *d_out, const float *d_out_sfc, const float *d_mem_out, float *d_mem_in,
                                       float *grads, void *stream)
{
    if (!h || !h->tr || slot < 0 || slot >= (int)h->tr->slots.size() || !x_main || !x_sfc || !rnn_mem || !x_denorm || !d_out || !d_out_sfc ||
        !d_mem_out || !d_mem_in || !grads) {
        csa_set_error_msg("csa_phys_train_backward: bad argument");
        return CSA_ERR_ARG;
    }
    PhysTrain *t = h->tr;
    PhysTrain::Slot &S = t->slots[slot];
    if (S.B != B || B <= 0) { csa_set_error_msg("csa_phys_train_backward: no pending forward of this batch size in the slot"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const PhysDev &d = t->d;
    const int nh = d.nh, L = PH_L, M = L * B, HDW = d.hdw, K1 = nh + 16, ns = t->nsplit, npart = pb_part_floats(nh);
    float *H2 = S.Hs2 + (size_t)B * nh;
    int rc;
    hipLaunchKernelGGL(phys_decode_bwd_kernel, dim3(B), dim3(PB_T), 0, s, d, B, S.HD, H2, x_sfc, rnn_mem, x_denorm, nxd, d_out, d_out_sfc, d_mem_out,
                       t->dHD, t->dlast, t->dpold, t->part);
    CSA_HIP_CHECK(hipGetLastError());
    // every reduction of this call is queued and runs as one launch at its end
    ReduceJobs &rq = t->rq;
    rq.n = 0; rq.nblk = 0;
    if ((rc = reduce_queue_add_2stage(rq, t->part, B, npart, t->m_dec, nullptr, t->rtmp, 32, s))) return rc;
    static const bool side_on = !getenv("CSA_PHYS_TRAIN_SIDE") || atoi(getenv("CSA_PHYS_TRAIN_SIDE")) != 0;
    hipStream_t sw = s;                                   // the stream of the weight-gradient GEMMs
    int nev = 0;
    if (side_on) {
        if (!t->side) {
            CSA_HIP_CHECK(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));
            for (hipEvent_t &e : t->ev) CSA_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        sw = t->side;
    }
    // the side stream picks up what the main stream has produced so far
    auto handoff = [&]() -> int {
        if (!side_on) return CSA_OK;
        CSA_HIP_CHECK(hipEventRecord(t->ev[nev], s));
        CSA_HIP_CHECK(hipStreamWaitEvent(sw, t->ev[nev], 0));
        ++nev;
        return CSA_OK;
    };
    float *pa = t->arena;
    // dW = A^T Bm as split-M partials (+ the column sums of A for the bias when ba is given), reduction queued
    auto wgrad = [&](const float *A, int lda, const float *Bm, int ldb, int Mr, int N1, int N2, const int *map, const int *ba,
                     const int *bb) -> int {
        float *cp = pa, *cb = nullptr;
        pa += (size_t)ns * N1 * N2;
        if (ba) { cb = pa; pa += (size_t)ns * N1; }
        if ((size_t)(pa - t->arena) > t->arena_floats) { csa_set_error_msg("csa_phys_train_backward: partial arena too small"); return CSA_ERR_ARG; }
        int r = ba ? launch_gemm_tn_partial_cs(A, lda, Bm, ldb, cp, cb, Mr, N1, N2, ns, sw) : launch_gemm_tn_partial(A, lda, Bm, ldb, cp, Mr, N1, N2, ns, sw);
        if (r) return r;
        if ((r = reduce_queue_add(rq, cp, ns, N1 * N2, map, nullptr))) return r;
        return ba ? reduce_queue_add(rq, cb, ns, N1, ba, bb) : CSA_OK;
    };
    // head GEMM
    if ((rc = launch_proj_gemm(t->dHD, t->wheadT, nullptr, t->dH2, M, nh, HDW, s))) return rc;
    hipLaunchKernelGGL(phys_add_last_kernel, dim3((B * nh + 255) / 256), dim3(256), 0, s, t->dH2 + (size_t)(L - 1) * B * nh, t->dlast, B * nh);
    CSA_HIP_CHECK(hipGetLastError());
    // rnn2 (downward)
    if ((rc = launch_bwd_rec_gru(nh, t->whh2Tp, S.GP2, S.Hs2, t->dH2, t->dhx2, B, L, 0, s))) return rc;
    if ((rc = handoff())) return rc;                      // dHD, GP2
    if ((rc = wgrad(t->dHD, HDW, H2, nh, M, HDW, nh, t->m_whead, t->m_bhead, nullptr))) return rc;
    if ((rc = wgrad(S.GP2, 4 * nh, S.H1, nh, M, 4 * nh, nh, t->m_wih2, t->m_b2a, t->m_b2b))) return rc;
    if ((rc = wgrad(S.GP2, 4 * nh, S.Hs2, nh, M, 4 * nh, nh, t->m_whh2, nullptr, nullptr))) return rc;
    if ((rc = launch_proj_gemm(S.GP2, t->wih2T, nullptr, t->dH1, M, nh, 4 * nh, s))) return rc;
    // rnn1 (upward): dH1 is in level order, the recurrence ran over the flipped axis
    if ((rc = launch_bwd_rec_gru(nh, t->whh1Tp, S.GP1, S.Hs1, t->dH1, t->dhx1, B, L, 1, s))) return rc;
    if ((rc = handoff())) return rc;                      // GP1
    if ((rc = wgrad(S.GP1, 4 * nh, S.X1, K1, M, 4 * nh, K1, t->m_wih1, t->m_b1a, t->m_b1b))) return rc;
    if ((rc = wgrad(S.GP1, 4 * nh, S.Hs1, nh, M, 4 * nh, nh, t->m_whh1, nullptr, nullptr))) return rc;
    if ((rc = launch_proj_gemm(S.GP1, t->wih1T, nullptr, t->dX1, M, K1, 4 * nh, s))) return rc;
    // mlp_initial, mlp_surface1, incoming memory (DPRE reuses dH1: rnn1's BPTT has consumed it)
    hipLaunchKernelGGL(phys_prep_bwd_kernel, dim3(B), dim3(128), 0, s, d, B, x_main, x_sfc, S.X1, S.hx, t->dX1, t->dhx1, t->dpold, t->dH1, t->XIN,
                       t->DHX, t->XS, d_mem_in);
    CSA_HIP_CHECK(hipGetLastError());
    if ((rc = handoff())) return rc;                      // DPRE, XIN, DHX, XS
    if ((rc = wgrad(t->dH1, nh, t->XIN, 32, M, nh, 32, t->m_init, t->m_initb, nullptr))) return rc;
    if ((rc = wgrad(t->DHX, nh, t->XS, 32, B, nh, 32, t->m_s1, t->m_s1b, nullptr))) return rc;
    if ((rc = launch_reduce_queue(rq, grads, sw))) return rc;
    if (side_on) {
        CSA_HIP_CHECK(hipEventRecord(t->ev[nev], sw));
        CSA_HIP_CHECK(hipStreamWaitEvent(s, t->ev[nev], 0));
    }
    S.B = 0;
    return CSA_OK;
}

__global__ void phys_sp_kernel(const float *__restrict__ xs, int nxs, float a, float bconst, float *__restrict__ sp, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sp[i] = xs[(size_t)i * nxs] * a + bconst;
}

// The reference trainer's loss on the physRNN outputs (rnn/utils.py:1203-1366 with rnn/metrics.py:142-163, 193-315: huber + energy +
// water closures, mp_mode 1 post-processing of the 5 -> 6 variables) and its analytic gradient: the loss kernels of the TBPTT trainer
// (train_misc.hip) on this model's scale factors.  Arguments as csa_train_loss; x_raw (Tw*B, 60, nxd) = inputs_denorm.
extern "C" int csa_phys_train_loss(csa_phys *h, int B, int Tw, int nxd, float w_energy, float w_water, const float *pred, const float *pred_sfc,
                                   const float *tgt, const float *tgt_sfc, const float *yto, const float *yto_sfc, const float *x_raw,
                                   const float *x_sfc_n, float *scalars, float *d_pred, float *d_pred_sfc, void *stream)
{
    if (!h || !h->tr || B <= 0 || B > h->max_batch || Tw <= 0 || Tw > (int)h->tr->slots.size() || nxd < 5 || !pred || !pred_sfc || !tgt || !tgt_sfc ||
        !yto || !yto_sfc || !x_raw || !x_sfc_n || !scalars) {
        csa_set_error_msg("csa_phys_train_loss: bad argument (the window may hold as many steps as the trainer has slots)");
        return CSA_ERR_ARG;
    }
    PhysTrain *t = h->tr;
    hipStream_t s = (hipStream_t)stream;
    const int N = B * Tw;
    DevModel m{};
    m.cfg.nlev = PH_L; m.cfg.nx = nxd; m.cfg.ny = 5; m.cfg.ny_sfc = 8; m.cfg.mp_mode = 1; m.cfg.nx_sfc = h->d.naux;
    m.yscale_lev = h->d.yscale_lev; m.yscale_sca = h->d.yscale_sca;
    hipLaunchKernelGGL(phys_sp_kernel, dim3((N + 255) / 256), dim3(256), 0, s, x_sfc_n, h->d.naux, h->d.xdiv_sca0, h->d.xmean_sca0, t->sp, N);
    CSA_HIP_CHECK(hipGetLastError());
    return launch_loss(m, h->d.hyai, h->d.hybi, B, Tw, w_energy, w_water, pred, pred_sfc, tgt, tgt_sfc, yto, yto_sfc, x_raw, t->sp, t->samp, t->ecoef,
                       scalars, d_pred, d_pred_sfc, s);
}

// Adam step on the flat parameter vector (torch.optim.Adam semantics, as csa_train_adam), then the kernel layouts re-packed
extern "C" int csa_phys_train_adam_step(csa_phys *h, const float *grads, float lr, float beta1, float beta2, float eps, float weight_decay,
                                        void *stream)
{
    if (!h || !h->tr || !grads) return CSA_ERR_ARG;
    PhysTrain *t = h->tr;
    int rc;
    ++t->adam_step;
    if ((rc = launch_adam(t->params, grads, t->adam_m, t->adam_v, t->nparam, lr, beta1, beta2, eps, t->adam_step, weight_decay, (hipStream_t)stream))) return rc;
    return pt_repack(t, (hipStream_t)stream);
}
