// head.hip -- output heads + wrapper post-processing + packing.  One workgroup per column.
//
// Reference semantics:
//   mlp_latent / mlp_output / output_prune / mlp_surface_output   rnn/models/models.py:547-560
//   de-normalisation + microphysics partition (mp_mode 1)          rnn/models/models.py:273-339,
//                                                                  rnn/save_wrapper_mem.py:470-483
//   packing (B,368[+nlev*nh_mem]) and NaN scrub                    rnn/save_wrapper_mem.py:485-497,539
//   tuple output + NaN scrub on out_lev                            rnn/utils.py:286-295
//
// HBM-bound: reads the rnn2 hidden sequence of its column (L*nh2 floats) once into LDS and the
// raw T/qliq/qice inputs, writes the packed row with level-contiguous (coalesced) stores.
#include "common.h"

#define HEAD_THREADS 256

__global__ __launch_bounds__(HEAD_THREADS) void head_kernel(
    DevModel m, int B, int mode, const float *__restrict__ H2, const float *__restrict__ x_raw,
    float *__restrict__ y0, float *__restrict__ y1, float *__restrict__ y2)
{
    extern __shared__ float smem[];
    const int L = m.cfg.nlev, nx = m.cfg.nx, ny = m.cfg.ny, nys = m.cfg.ny_sfc;
    const int nh2 = m.cfg.nh2, nm = m.cfg.nh_mem, ldh = nh2 + 1;
    float *hs = smem;                       // (L, nh2+1)
    float *zs = hs + L * ldh;               // (L, nm)
    float *os = zs + L * (nm > 0 ? nm : 1); // (L, ny)
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool legacy = m.cfg.legacy != 0;

    for (int idx = tid; idx < L * nh2; idx += HEAD_THREADS) {
        const int l = idx / nh2, k = idx - l * nh2;
        hs[l * ldh + k] = H2[((size_t)l * B + b) * nh2 + k];
    }
    __syncthreads();

    // ---- latent memory -----------------------------------------------------------------------
    if (nm > 0) {
        float *mem_out = mode == HEAD_PACKED ? y0 + (size_t)b * (6 * L + nys + L * nm) + 6 * L + nys
                                             : y2;
        for (int idx = tid; idx < L * nm; idx += HEAD_THREADS) {
            const int l = idx / nm, j = idx - l * nm;
            float a = m.lat_b[j];
            const float *hr = hs + l * ldh;
            for (int k = 0; k < nh2; ++k) a += hr[k] * m.lat_wt[k * nm + j];
            zs[idx] = a;
            float v = a;
            if (mode == HEAD_PACKED) {
                if (m.cfg.scrub_out_nan && isnan(v)) v = 0.0f;
                // packed rows carry the model's memory block verbatim: legacy (L,nm) in sequence order
                mem_out[(legacy ? (L - 1 - l) : l) * nm + j] = v;
            } else if (legacy) {
                mem_out[((size_t)b * L + (L - 1 - l)) * nm + j] = v;
            } else {
                mem_out[((size_t)l * B + b) * nm + j] = v;
            }
        }
        __syncthreads();
    }
    // ---- level outputs (normalised) -----------------------------------------------------------
    for (int idx = tid; idx < L * ny; idx += HEAD_THREADS) {
        const int l = idx / ny, v = idx - l * ny;
        float a = m.out_b[v];
        if (nm > 0) {
            for (int j = 0; j < nm; ++j) a += zs[l * nm + j] * m.out_w[v * nm + j];
        } else {
            const float *hr = hs + l * ldh;
            for (int k = 0; k < nh2; ++k) a += hr[k] * m.out_w[v * nh2 + k];
        }
        if (m.cfg.output_prune && l < 12 && v >= 1) a = 0.0f;
        os[idx] = a;
    }
    __syncthreads();

    // ---- de-normalise, microphysics, pack -------------------------------------------------------
    const bool post = (mode != HEAD_RAW) && m.cfg.mp_mode == 1;
    for (int l = tid; l < L; l += HEAD_THREADS) {
        const float *o = os + l * ny;
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
            const float *ys = m.yscale_lev + l * ny;
            const float *xr = x_raw + ((size_t)b * L + l) * nx;
            const float dT = o[0] / ys[0], dqv = o[1] / ys[1], dqn = o[2] / ys[2];
            const float du = o[3] / ys[3], dv = o[4] / ys[4];
            const float T_old = xr[0], ql = xr[2], qi = xr[3];
            const float T_new = T_old + dT * 1200.0f;
            float lf = (T_new - 253.16f) * 0.05f;
            if (!isnan(lf)) lf = fminf(fmaxf(lf, 0.0f), 1.0f);
            const float qn_new = (ql + qi) + dqn * 1200.0f;
            const float dql = (lf * qn_new - ql) * 0.0008333333333333334f;
            const float dqi = ((1.0f - lf) * qn_new - qi) * 0.0008333333333333334f;
            float vals[6] = {dT, dqv, dql, dqi, du, dv};
            if (mode == HEAD_PACKED) {
                float *y = y0 + (size_t)b * (6 * L + nys + L * nm);
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
    // ---- surface outputs -------------------------------------------------------------------------
    for (int v = tid; v < nys; v += HEAD_THREADS) {
        float a = m.sfo_b[v];
        const float *hr = hs + (L - 1) * ldh;     // last_h of the downward RNN
        for (int k = 0; k < nh2; ++k) a += hr[k] * m.sfo_w[v * nh2 + k];
        if (mode == HEAD_PACKED) {
            a = a / m.yscale_sca[v];
            if (m.cfg.scrub_out_nan && isnan(a)) a = 0.0f;
            y0[(size_t)b * (6 * L + nys + L * nm) + 6 * L + v] = a;
        } else {
            if (mode == HEAD_TUPLE && m.cfg.mp_mode != 0) a = a / m.yscale_sca[v];
            y1[(size_t)b * nys + v] = a;
        }
    }
}

int launch_head(const DevModel &m, int B, int mode, const float *H2, const float *x_main_raw,
                float *y0, float *y1, float *y2, hipStream_t s)
{
    const int L = m.cfg.nlev, nm = m.cfg.nh_mem;
    const size_t shm = sizeof(float) * ((size_t)L * (m.cfg.nh2 + 1) + (size_t)L * (nm > 0 ? nm : 1) + (size_t)L * m.cfg.ny);
    if (shm > 64 * 1024) {
        csa_set_error_msg("head: LDS footprint exceeds 64 KB");
        return CSA_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(head_kernel, dim3(B), dim3(HEAD_THREADS), shm, s, m, B, mode, H2, x_main_raw, y0, y1, y2);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
