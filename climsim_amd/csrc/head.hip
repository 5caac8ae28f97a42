// head.hip -- output heads + wrapper post-processing + packing.  One workgroup per column.
//
// Reference semantics:
//   mlp_latent / mlp_output / output_prune / mlp_surface_output   rnn/models/models.py:547-560
//   de-normalisation + microphysics partition (mp_mode 1,-1,-2)    rnn/models/models.py:273-339,
//                                                                  rnn/save_wrapper_mem.py:470-483
//   packing (B,368[+nlev*nh_mem]) and NaN scrub                    rnn/save_wrapper_mem.py:485-497,539
//   tuple output + NaN scrub on out_lev                            rnn/utils.py:286-295
//
// HBM-bound: reads the rnn2 hidden sequence of its column (L*nh2 floats) once into LDS and the
// raw T/qliq/qice inputs, writes the packed row with level-contiguous (coalesced) stores.
#include "common.h"
#include "rh_to_q.h"
#include <cstdlib>

#define HEAD_THREADS 256

// De-normalisation + microphysics partition of ONE level (models.py:273-339; wrapper twin save_wrapper_mem.py:470-483):
// o = the ny normalised model outputs of level l, xr = the raw level inputs of that cell -> vals = [dT,dqv,dqliq,dqice,du,dv].
__device__ __forceinline__ void postprocess_level(const DevModel &m, int l, int b, const float *o, const float *xr,
                                                  const float *x_sfc_raw, float *vals)
{
    const int ny = m.cfg.ny, mp = m.cfg.mp_mode;
    const float *ys = m.yscale_lev + l * ny;
    const int nxr = m.cfg.nx - (m.cfg.q_input_mode == 1);
    // mp_mode 1: [dT,dqv,dqn,du,dv]; -1: [dT,dqv,dqn,liq_frac,du,dv]; -2: [dT,dqtot,cld_frac,liq_frac,du,dv]
    const int iu = mp == 1 ? 3 : 4;
    const float dT = o[0] / ys[0];
    float dqv = o[1] / ys[1], dqn = o[2] / ys[2];
    const float du = o[iu] / ys[iu], dv = o[iu + 1] / ys[iu + 1];
    const float T_old = xr[0], ql = xr[2], qi = xr[3];
    if (mp == -2) {
        // models.py:286-301: total-water tendency + cloud fraction of total water -> dqv, dqn.
        // q_old is the LAST raw level input (the appended specific humidity when include_q_input)
        float qv_old;
        if (m.cfg.q_input_mode == 1) {
            const float pres = m.hyam[l] * 100000.0f + x_sfc_raw[(size_t)b * m.cfg.nx_sfc] * m.hybm[l];
            qv_old = prep_rh_to_q(xr[1], xr[0], pres);
        } else {
            qv_old = xr[nxr - 1];
        }
        float cf = dqn * dqn;
        cf = cf * cf;
        if (!isnan(cf)) cf = fminf(fmaxf(cf, 0.0f), 1.0f);
        const float qn_old = ql + qi;
        const float qtot_new = (qn_old + qv_old) + dqv * 1200.0f;
        const float qv_new = (1.0f - cf) * qtot_new, qn_new2 = cf * qtot_new;
        dqv = (qv_new - qv_old) * 0.0008333333333333334f;
        dqn = (qn_new2 - qn_old) * 0.0008333333333333334f;
    }
    const float T_new = T_old + dT * 1200.0f;
    float lf;
    if (mp == 1) {
        lf = (T_new - 253.16f) * 0.05f;
        if (!isnan(lf)) lf = fminf(fmaxf(lf, 0.0f), 1.0f);
    } else {
        lf = o[3] / ys[3];      // models.py:319: the clamped value is overwritten by the raw prediction
    }
    const float qn_new = (ql + qi) + dqn * 1200.0f;
    const float dql = (lf * qn_new - ql) * 0.0008333333333333334f;
    const float dqi = ((1.0f - lf) * qn_new - qi) * 0.0008333333333333334f;
    vals[0] = dT; vals[1] = dqv; vals[2] = dql; vals[3] = dqi; vals[4] = du; vals[5] = dv;
}

// First stage of the head: Z(l, j) = b1[j] + sum_k H2(l, k) * W1t(k, j), j < n1 <= 16, where (W1t, n1) is
// mlp_latent (n1 = nh_mem) for the memory models or mlp_output itself (n1 = ny) for the stateless one: a (60 x NH2) x
// (NH2 x 16) product per column, done with 16x16x4 fp32 MFMAs on the hidden rows staged in LDS (see below).
template <int NH2>
__global__ __launch_bounds__(HEAD_THREADS) void head_kernel(
    DevModel m, int B, int mode, const float *__restrict__ H2, const float *__restrict__ x_raw,
    const float *__restrict__ x_sfc_raw, float *__restrict__ y0, float *__restrict__ y1, float *__restrict__ y2)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = m.cfg.nlev, nx = m.cfg.nx, ny = m.cfg.ny, nys = m.cfg.ny_sfc;
    const int nm = m.cfg.nh_mem;
    constexpr int ldh = NH2 + 4;            // float4-aligned rows, 16-B skew between rows
    // a column's levels are cut into gridDim.y slices (one workgroup each): more, shorter workgroups in flight
    const int Ls = (L + gridDim.y - 1) / gridDim.y, l0 = blockIdx.y * Ls, nl = min(L, l0 + Ls) - l0;
    if (nl <= 0) return;
    float *hs = smem;                        // (Ls, ldh)   rows of this slice
    float *zs = hs + Ls * ldh;               // (Ls, nm)
    float *os = zs + Ls * (nm > 0 ? nm : 1); // (Ls, ny)
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool legacy = m.cfg.legacy != 0;
    const int W = 6 * L + nys + L * nm;     // packed row width

    for (int idx = tid; idx < nl * (NH2 / 4); idx += HEAD_THREADS) {
        const int ll = idx / (NH2 / 4), k4 = idx - ll * (NH2 / 4);
        *(f32x4 *)&hs[ll * ldh + 4 * k4] = *(const f32x4 *)&H2[((size_t)(l0 + ll) * B + b) * NH2 + 4 * k4];
    }
    // ---- first stage on the matrix pipe: Z (L x n1) = H2 (L x NH2) . W1t (NH2 x n1) + b1, n1 <= 16 -------------------------------
    // v_mfma_f32_16x16x4_f32: wave w takes the 16-level row block w (L = 60: four blocks, one per wave).  A = hidden rows from
    // LDS (lane: row lane%16, k lane/16), B = the weight matrix itself, (NH2, n1) row-major: for n1 = 16 lane `lane` of k-step
    // ks reads w1t[64*ks + lane] -- fully coalesced, 32 registers per lane instead of the NH2 a thread-per-output layout keeps
    // (which limited the kernel to 3 workgroups per CU and re-read 128 KB of weights per column from L2).
    const int n1 = nm > 0 ? nm : ny;
    const float *w1t = nm > 0 ? m.lat_wt : m.out_wt;   // (NH2, n1)
    const int lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    float wB[NH2 / 4];
#pragma unroll
    for (int ks = 0; ks < NH2 / 4; ++ks) wB[ks] = col < n1 ? w1t[(4 * ks + kq) * n1 + col] : 0.0f;
    const float b1 = col < n1 ? (nm > 0 ? m.lat_b[col] : m.out_b[col]) : 0.0f;
    __syncthreads();
    {
        float *mem_out = mode == HEAD_PACKED ? y0 + (size_t)b * W + 6 * L + nys : y2;
        for (int r0 = 16 * wave; r0 < nl; r0 += 16 * (HEAD_THREADS / 64)) {
            const float *arow = hs + min(r0 + col, nl - 1) * ldh + kq;     // rows beyond the slice: recompute the last one, discard
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
            for (int ks = 0; ks < NH2 / 4; ks += 2) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * ks], wB[ks], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * ks + 4], wB[ks + 1], d1, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                                   // D: row 4*(lane/16) + i, column lane%16
                const int ll = r0 + 4 * kq + i, l = l0 + ll;
                const float a = (d0[i] + d1[i]) + b1;
                if (ll >= nl || col >= n1) continue;
                if (nm > 0) {
                    zs[ll * nm + col] = a;
                    float v = a;
                    if (mode == HEAD_PACKED) {
                        if (m.cfg.scrub_out_nan && isnan(v)) v = 0.0f;
                        // packed rows carry the model's memory block verbatim: legacy (L,nm), sequence order
                        mem_out[(legacy ? (L - 1 - l) : l) * nm + col] = v;
                    } else if (legacy) {
                        mem_out[((size_t)b * L + (L - 1 - l)) * nm + col] = v;
                    } else {
                        mem_out[((size_t)l * (m.mem_B > 0 ? m.mem_B : B) + m.mem_off + b) * nm + col] = v;
                    }
                } else {
                    os[ll * ny + col] = (m.cfg.output_prune && l < 12 && col >= 1) ? 0.0f : a;
                }
            }
        }
    }
    __syncthreads();
    // ---- second stage for the memory models: out = mlp_output(latent) ---------------------------
    if (nm > 0) {
        for (int idx = tid; idx < nl * ny; idx += HEAD_THREADS) {
            const int ll = idx / ny, v = idx - ll * ny, l = l0 + ll;
            float a = m.out_b[v];
            if (nm == 16) {        // compile-time trip count: the 16 weight loads are issued together (run-time bound: one s_waitcnt each)
                float w[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) w[q] = m.out_w[v * 16 + q];
#pragma unroll
                for (int q = 0; q < 16; ++q) a += zs[ll * 16 + q] * w[q];
            } else
                for (int q = 0; q < nm; ++q) a += zs[ll * nm + q] * m.out_w[v * nm + q];
            if (m.cfg.output_prune && l < 12 && v >= 1) a = 0.0f;
            os[idx] = a;
        }
        __syncthreads();
    }

    // ---- de-normalise, microphysics, pack -------------------------------------------------------
    const bool post = (mode != HEAD_RAW) && m.cfg.mp_mode != 0;
    for (int ll = tid; ll < nl; ll += HEAD_THREADS) {
        const int l = l0 + ll;
        const float *o = os + ll * ny;
        if (mode == HEAD_RAW || m.cfg.mp_mode == 0) {
            // RAW: model-level output; TUPLE with mp_mode 0 returns the UN-denormalised outputs
            // (models.py:278-279).  PACKED with mp_mode 0 is rejected on the host.
            float *dst = y0 + ((size_t)b * L + l) * ny;
            for (int v = 0; v < ny; ++v) {
                float val = o[v];
                if (mode == HEAD_TUPLE && isnan(val)) val = 0.0f;
                dst[v] = val;
            }
        } else if (post) {
            const int nxr = nx - (m.cfg.q_input_mode == 1);
            float vals[6];
            postprocess_level(m, l, b, o, x_raw + ((size_t)b * L + l) * nxr, x_sfc_raw, vals);
            if (mode == HEAD_PACKED) {
                float *y = y0 + (size_t)b * W;
#pragma unroll
                for (int v = 0; v < 6; ++v) {
                    float val = vals[v];
                    if (m.cfg.scrub_out_nan && isnan(val)) val = 0.0f;
                    y[v * L + l] = val;
                }
            } else {
                float *y = y0 + ((size_t)b * L + l) * 6;
#pragma unroll
                for (int v = 0; v < 6; ++v) y[v] = isnan(vals[v]) ? 0.0f : vals[v];
            }
        }
    }
    // ---- surface outputs: the slice that holds the last level; the SECOND wave (the first is busy with the loop above),
    // 8 lanes per output
    if (l0 + nl == L && tid >= 64 && tid < 128) {
        const int v = (tid - 64) >> 3, part = tid & 7;
        float a = 0.0f;
        if (v < nys) {
            const float *hr = hs + (nl - 1) * ldh;    // last_h of the downward RNN
            for (int k = part; k < NH2; k += 8) a += hr[k] * m.sfo_w[v * NH2 + k];
        }
        a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
        if (v < nys && part == 0) {
            a += m.sfo_b[v];
            if (mode == HEAD_PACKED) {
                a = a / m.yscale_sca[v];
                if (m.cfg.scrub_out_nan && isnan(a)) a = 0.0f;
                y0[(size_t)b * W + 6 * L + v] = a;
            } else {
                if (mode == HEAD_TUPLE && m.cfg.mp_mode != 0) a = a / m.yscale_sca[v];
                y1[(size_t)b * nys + v] = a;
            }
        }
    }
}

int launch_head(const DevModel &m, int B, int mode, const float *H2, const float *x_main_raw,
                const float *x_sfc_raw, float *y0, float *y1, float *y2, hipStream_t s)
{
    const int L = m.cfg.nlev, nm = m.cfg.nh_mem;
    // level slices per column (CSA_HEAD_LSPLIT, default 2): measured at 384 / 2,700 columns, see DESIGN.md section 4.3
    static const int lsplit = getenv("CSA_HEAD_LSPLIT") ? atoi(getenv("CSA_HEAD_LSPLIT")) : 2;
    const int S = lsplit < 1 ? 1 : (lsplit > 4 ? 4 : lsplit), Ls = (L + S - 1) / S;
    const size_t shm = sizeof(float) * ((size_t)Ls * (m.cfg.nh2 + 4) + (size_t)Ls * (nm > 0 ? nm : 1) + (size_t)Ls * m.cfg.ny);
    if (shm > 64 * 1024 || m.cfg.ny_sfc > 8 || (nm > 0 ? nm : m.cfg.ny) > 16) {
        csa_set_error_msg("head: unsupported sizes (LDS > 64 KB, ny_sfc > 8 or first-stage width > 16)");
        return CSA_ERR_UNSUPPORTED;
    }
    switch (m.cfg.nh2) {
    case 64:  hipLaunchKernelGGL(head_kernel<64>, dim3(B, S), dim3(HEAD_THREADS), shm, s, m, B, mode, H2, x_main_raw, x_sfc_raw, y0, y1, y2); break;
    case 96:  hipLaunchKernelGGL(head_kernel<96>, dim3(B, S), dim3(HEAD_THREADS), shm, s, m, B, mode, H2, x_main_raw, x_sfc_raw, y0, y1, y2); break;
    case 128: hipLaunchKernelGGL(head_kernel<128>, dim3(B, S), dim3(HEAD_THREADS), shm, s, m, B, mode, H2, x_main_raw, x_sfc_raw, y0, y1, y2); break;
    case 144: hipLaunchKernelGGL(head_kernel<144>, dim3(B, S), dim3(HEAD_THREADS), shm, s, m, B, mode, H2, x_main_raw, x_sfc_raw, y0, y1, y2); break;
    default:
        csa_set_error_msg("head: hidden size must be 64, 96, 128 or 144");
        return CSA_ERR_UNSUPPORTED;
    }
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// ---- RNN_autoreg.postprocessing(out, out_sfc, x_denorm) as its own entry (models.py:273-339) ---------------------------------
// One thread per (column, level); HBM-bound: reads ny + 4 floats, writes 6 per cell.  mp_mode 0 returns its inputs unchanged
// (:278-279), so the host mirror does not launch anything for it.
__global__ __launch_bounds__(256) void post_kernel(DevModel m, int B, const float *__restrict__ out, const float *__restrict__ out_sfc,
                                                   const float *__restrict__ x_denorm, int nxd, float *__restrict__ out6,
                                                   float *__restrict__ out_sfc_d)
{
    const int L = m.cfg.nlev, ny = m.cfg.ny, nys = m.cfg.ny_sfc;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * L) {
        const int b = i / L, l = i - b * L;
        float o[8], vals[6];
        for (int v = 0; v < ny; ++v) o[v] = out[(size_t)i * ny + v];
        // x_denorm carries at least T, (RH), qliq, qice; mp_mode -2 reads its LAST column as the old specific humidity
        const float *xr = x_denorm + (size_t)i * nxd;
        DevModel mm = m;
        mm.cfg.q_input_mode = 0;                 // the caller passes the tensor the model saw (q already appended)
        mm.cfg.nx = nxd;
        postprocess_level(mm, l, b, o, xr, nullptr, vals);
        for (int v = 0; v < 6; ++v) out6[(size_t)i * 6 + v] = vals[v];
    }
    if (i < B * nys) out_sfc_d[i] = out_sfc[i] / m.yscale_sca[i % nys];
}

int launch_postprocess(const DevModel &m, int B, const float *out, const float *out_sfc, const float *x_denorm, int nxd,
                       float *out6, float *out_sfc_d, hipStream_t s)
{
    if (m.cfg.ny > 8 || nxd < 4) { csa_set_error_msg("postprocess: ny <= 8 and x_denorm with at least 4 columns required"); return CSA_ERR_ARG; }
    const int n = B * m.cfg.nlev;
    hipLaunchKernelGGL(post_kernel, dim3((n + 255) / 256), dim3(256), 0, s, m, B, out, out_sfc, x_denorm, nxd, out6, out_sfc_d);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// ---- packed row of the stateful + AR-noise wrapper (save_wrapper_mem.py:640-681 mp_postprocessing with eps_prev) ------------------
// yout (B, 368 + nlev*nh_mem + nlev*nh) = [dT,dqv (120) | dqliq (60) | dqice (60) | du,dv (120) | sfc (8) | memory of the column
// (nlev,nh_mem) | eps of the column (nlev,nh)], NaN -> 0 (:717).  Inputs are the tuple outputs of the model call: out6 (B,L,6),
// out_sfc (B,8), mem (L,B,nm) level-major, eps (L,B,nh) level-major.  One workgroup per column; pure data movement.
__global__ __launch_bounds__(256) void pack_ar_kernel(int B, int L, int nys, int nm, int nh, const float *__restrict__ out6,
                                                      const float *__restrict__ out_sfc, const float *__restrict__ mem,
                                                      const float *__restrict__ eps, float *__restrict__ yout)
{
    const int b = blockIdx.x, tid = threadIdx.x, W = 6 * L + nys + L * nm + L * nh;
    float *y = yout + (size_t)b * W;
    auto scrub = [](float v) { return isnan(v) ? 0.0f : v; };
    for (int i = tid; i < 6 * L; i += 256) { const int v = i / L, l = i - v * L; y[i] = scrub(out6[((size_t)b * L + l) * 6 + v]); }
    for (int i = tid; i < nys; i += 256) y[6 * L + i] = scrub(out_sfc[(size_t)b * nys + i]);
    for (int i = tid; i < L * nm; i += 256) { const int l = i / nm, k = i - l * nm; y[6 * L + nys + i] = scrub(mem[((size_t)l * B + b) * nm + k]); }
    for (int i = tid; i < L * nh; i += 256) { const int l = i / nh, k = i - l * nh; y[6 * L + nys + L * nm + i] = scrub(eps[((size_t)l * B + b) * nh + k]); }
}
// (B, L, n) batch-first -> (L, B, n) level-major (the wrapper's torch.transpose(eps_prev, 0, 1), :693)
__global__ __launch_bounds__(256) void to_level_major_kernel(int B, int L, int n, const float *__restrict__ src, float *__restrict__ dst)
{
    const long total = (long)B * L * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % n);
        const long r = i / n;
        const int l = (int)(r % L), b = (int)(r / L);
        dst[((size_t)l * B + b) * n + k] = src[i];
    }
}

int launch_to_level_major(int B, int L, int n, const float *src, float *dst, hipStream_t s)
{
    const long total = (long)B * L * n, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(to_level_major_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, B, L, n, src, dst);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

int launch_pack_ar(int B, int L, int nys, int nm, int nh, const float *out6, const float *out_sfc, const float *mem,
                   const float *eps, float *yout, hipStream_t s)
{
    hipLaunchKernelGGL(pack_ar_kernel, dim3(B), dim3(256), 0, s, B, L, nys, nm, nh, out6, out_sfc, mem, eps, yout);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
