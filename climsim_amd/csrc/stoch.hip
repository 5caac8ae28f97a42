// stoch.hip -- the reference's stochastic recurrent layers.
//
//   MyStochasticGRULayer5   rnn/models_torch_kernels.py:834-891; its GPU path is the repository's only native
//                           code: reparam_forward_kernel :35-53, gru_gate_forward_kernel :55-82 and the C++
//                           sequence loop :129-174 (two cuBLAS GEMMs + two elementwise launches PER LEVEL).
//   MyStochasticLSTMLayer4  rnn/models_torch_kernels.py:1474-1531.
//
// MI355X version: the hoisted input projection (x @ W_ih for all levels, the reference does the same at
// :858-862) is one fp32-MFMA GEMM; everything level-recurrent is ONE launch with all recurrent weights
// stationary in registers (GRU5: W_enc 128x256 + W_zh 128x384 = 320 KB; LSTM4: the hidden half of W_enc,
// 128x640 = 320 KB; 160 weights per lane of a 512-thread workgroup), two columns per workgroup advanced with
// packed FMAs, reparameterisation / gates fused in, LDS-only barriers -- i.e. the 240 launches + 120 GEMMs
// of the reference's sequence loop become 2 launches.  eps (T,B,H) is an explicit input (the reference
// draws it with torch.randn at the top of forward).
#include "common.h"
#include "stoch.h"
#include <vector>

#define PK_FMA_LO(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(h))
#define PK_FMA_HI(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(h))
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ float sq_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    return v;
}
// Quad sum of BOTH column components, then pick this lane's column.  Never write
// `col ? sq_sum(a.y) : sq_sum(a.x)`: C++ evaluates only the selected operand, the divergent condition becomes
// control flow, and the DPP adds then run with half the quad masked off (they read 0 from inactive lanes).
__device__ __forceinline__ float sq_pick(f32x2 a, int col)
{
    const float sx = sq_sum(a.x), sy = sq_sum(a.y);
    return col ? sy : sx;
}
__device__ __forceinline__ float s_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float s_tanh(float x)
{
    const float t = fminf(__builtin_amdgcn_exp2f(-2.88539008177792681f * x), 1e30f);
    return (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
}
__device__ __forceinline__ float s_exp(float x) { return __builtin_amdgcn_exp2f(1.44269504088896341f * x); }

// matvec over this lane's k-quarter: acc[r] += w[r][*] * v[*], v = (k,col) pairs in LDS
template <int R, int KC>
__device__ __forceinline__ void quarter_matvec(const f32x2 (&w)[R][KC / 2], const float *vbuf, f32x2 (&acc)[R])
{
    const f32x4 *vp = (const f32x4 *)vbuf;
#pragma unroll
    for (int j = 0; j < KC / 2; ++j) {
        const f32x4 hv = vp[j];
        const f32x2 ha = {hv.x, hv.y}, hb = {hv.z, hv.w};
#pragma unroll
        for (int r = 0; r < R; ++r) PK_FMA_LO(acc[r], w[r][j], ha);
#pragma unroll
        for (int r = 0; r < R; ++r) PK_FMA_HI(acc[r], w[r][j], hb);
    }
}

// Packed layout: float4 number i of thread tid at Wp4[i*NT + tid]; thread-local order idx = r*KC + kk
// <-> row r of unit u = tid>>2, k = (tid&3)*KC + kk.
template <int R, int KC>
__device__ __forceinline__ void load_rows(const f32x4 *Wp4, int NT, int tid, f32x2 (&w)[R][KC / 2])
{
#pragma unroll
    for (int i = 0; i < R * KC / 4; ++i) {
        const f32x4 v = Wp4[(size_t)i * NT + tid];
        const int r = (4 * i) / KC, kk = (4 * i) % KC;
        w[r][kk / 2] = f32x2{v.x, v.y};
        w[r][kk / 2 + 1] = f32x2{v.z, v.w};
    }
}

// ------------------------------------------------------------------------------------------------
// GRU5: per level  pred = h W_enc -> (mean, logvar);  z = mean + eps*exp(logvar/2);  zr = z W_zh (+b);
//   r = sig(r_u + z_r); zg = sig(zg_u + z_z); n = tanh(n_u + r*z_n); h' = n + zg*(h - n).
// XP (T,B,3H) = x W_ih (+b_ih), gate-major.
// TRAIN additionally saves what BPTT needs (stoch_bwd.hip): (r, zg, n) IN PLACE over the projections XP, z_n / z / exp(logvar/2)
// into ZN / Zs / EX (T,B,H) and h_{t-1} into Hseq (T+1 slots, slot 0 = h0) -- the tensors the reference's C++ loop keeps
// (all_r, all_zg, all_n, all_zn, all_z, all_exp, all_h; models_torch_kernels.py:141-174).
template <int NH, bool TRAIN = false>
__global__ __launch_bounds__(NH * 4, 2) void stoch_gru5_kernel(
    const f32x4 *__restrict__ Wenc4, const f32x4 *__restrict__ Wzh4, const float *__restrict__ bzh,
    float *__restrict__ XP, const float *__restrict__ eps, const float *__restrict__ h0,
    float *__restrict__ out, int B, int T, float *__restrict__ ZN = nullptr, float *__restrict__ Zs = nullptr,
    float *__restrict__ EX = nullptr, float *__restrict__ Hseq = nullptr)
{
    constexpr int NT = NH * 4, KC = NH / 4, CH = 2 * KC + 4;
    __shared__ __attribute__((aligned(16))) float hbuf[4 * CH];
    __shared__ __attribute__((aligned(16))) float zbuf[4 * CH];
    const int tid = threadIdx.x, u = tid >> 2, p = tid & 3, col = p & 1;
    int b = 2 * blockIdx.x + col;
    const bool valid = b < B;
    if (!valid) b = B - 1;
    f32x2 we[2][KC / 2], wz[3][KC / 2];
    load_rows<2, KC>(Wenc4, NT, tid, we);
    load_rows<3, KC>(Wzh4, NT, tid, wz);
    float bz[3] = {0.f, 0.f, 0.f};
    if (bzh) { bz[0] = bzh[u]; bz[1] = bzh[NH + u]; bz[2] = bzh[2 * NH + u]; }
    float h = h0[(size_t)b * NH + u];
    const int slot = 2 * u + col + 4 * (u / KC);
    if (p < 2) hbuf[slot] = h;
    if (TRAIN && p < 2 && valid) Hseq[(size_t)b * NH + u] = h;
    asm volatile("" : "+v"(h));
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * B + b;
        const float xr = XP[row * 3 * NH + u], xz = XP[row * 3 * NH + NH + u], xn = XP[row * 3 * NH + 2 * NH + u];
        const float e = eps[row * NH + u];
        // phase A: (mean, logvar) = h W_enc
        f32x2 a2[2] = {{0.f, 0.f}, {0.f, 0.f}};
        quarter_matvec<2, KC>(we, hbuf + p * CH, a2);
        const float mean = sq_pick(a2[0], col);
        const float logv = sq_pick(a2[1], col);
        const float ex = s_exp(0.5f * logv);
        const float z = mean + e * ex;
        if (p < 2) zbuf[slot] = z;
        LDS_BARRIER();
        // phase B: (z_r, z_z, z_n) = z W_zh (+ b_zh)
        f32x2 a3[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        quarter_matvec<3, KC>(wz, zbuf + p * CH, a3);
        const float z_r = sq_pick(a3[0], col) + bz[0];
        const float z_z = sq_pick(a3[1], col) + bz[1];
        const float z_n = sq_pick(a3[2], col) + bz[2];
        const float r = s_sigmoid(xr + z_r);
        const float zg = s_sigmoid(xz + z_z);
        const float n = s_tanh(xn + r * z_n);
        h = n + zg * (h - n);
        if (p < 2) {
            hbuf[slot] = h;
            if (valid) out[row * NH + u] = h;
            if (TRAIN && valid) {
                XP[row * 3 * NH + u] = r; XP[row * 3 * NH + NH + u] = zg; XP[row * 3 * NH + 2 * NH + u] = n;
                ZN[row * NH + u] = z_n; Zs[row * NH + u] = z; EX[row * NH + u] = ex;
                Hseq[((size_t)(t + 1) * B + b) * NH + u] = h;
            }
        }
        LDS_BARRIER();
    }
}

// ------------------------------------------------------------------------------------------------
// LSTM4: per level  yy = [x, h] W_enc -> (mean, logvar, i, f, g);  o = sig(mean + eps*exp(logvar/2));
//   c = sig(f) c + sig(i) tanh(g);  h = o tanh(c).     XP (T,B,5H) = x W_enc[:nx], gate-major.
// TRAIN: the activated values [o, exp(logvar/2), sig(i), sig(f), tanh(g)] overwrite the projections XP in place, h_{t-1} and
// c_{t-1} go to Hseq / Cseq (T+1 slots, slot 0 = the initial state).
template <int NH, bool TRAIN = false>
__global__ __launch_bounds__(NH * 4, 2) void stoch_lstm4_kernel(
    const f32x4 *__restrict__ Wh4, float *__restrict__ XP, const float *__restrict__ eps,
    const float *__restrict__ h0, const float *__restrict__ c0, float *__restrict__ out,
    float *__restrict__ hT, float *__restrict__ cT, int B, int T, float *__restrict__ Hseq = nullptr,
    float *__restrict__ Cseq = nullptr)
{
    constexpr int NT = NH * 4, KC = NH / 4, CH = 2 * KC + 4;
    __shared__ __attribute__((aligned(16))) float hbuf[2][4 * CH];
    const int tid = threadIdx.x, u = tid >> 2, p = tid & 3, col = p & 1;
    int b = 2 * blockIdx.x + col;
    const bool valid = b < B;
    if (!valid) b = B - 1;
    f32x2 w[5][KC / 2];
    load_rows<5, KC>(Wh4, NT, tid, w);
    float h = h0[(size_t)b * NH + u], c = c0[(size_t)b * NH + u];
    const int slot = 2 * u + col + 4 * (u / KC);
    if (p < 2) hbuf[0][slot] = h;
    if (TRAIN && p < 2 && valid) { Hseq[(size_t)b * NH + u] = h; Cseq[(size_t)b * NH + u] = c; }
    asm volatile("" : "+v"(h), "+v"(c));
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * B + b;
        float xp[5];
#pragma unroll
        for (int g = 0; g < 5; ++g) xp[g] = XP[row * 5 * NH + g * NH + u];
        const float e = eps[row * NH + u];
        f32x2 a[5] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        quarter_matvec<5, KC>(w, hbuf[t & 1] + p * CH, a);
        float s[5];
#pragma unroll
        for (int g = 0; g < 5; ++g) s[g] = sq_pick(a[g], col) + xp[g];
        const float ex = s_exp(0.5f * s[1]);
        const float o = s_sigmoid(s[0] + e * ex), ig = s_sigmoid(s[2]), fg = s_sigmoid(s[3]), gg = s_tanh(s[4]);
        c = fg * c + ig * gg;
        h = o * s_tanh(c);
        if (p < 2) {
            hbuf[(t & 1) ^ 1][slot] = h;
            if (valid) out[row * NH + u] = h;
            if (TRAIN && valid) {
                float *a = XP + row * 5 * NH + u;
                a[0] = o; a[NH] = ex; a[2 * NH] = ig; a[3 * NH] = fg; a[4 * NH] = gg;
                Hseq[((size_t)(t + 1) * B + b) * NH + u] = h;
                Cseq[((size_t)(t + 1) * B + b) * NH + u] = c;
            }
        }
        LDS_BARRIER();
    }
    if (p < 2 && valid) {
        if (hT) hT[(size_t)b * NH + u] = h;
        if (cT) cT[(size_t)b * NH + u] = c;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------
// pack R recurrent rows per unit from an (in=nh, out=ncols) matrix W (reference layout): row r of unit u is
// column (r*nh + u) of W
void stoch_pack_rows(int nh, int R, const float *W, int ncols, int col0, float *packed)
{
    const int NT = nh * 4, KC = nh / 4;
    for (int tid = 0; tid < NT; ++tid) {
        const int u = tid >> 2, p = tid & 3;
        for (int idx = 0; idx < R * KC; ++idx) {
            const int r = idx / KC, kk = idx % KC, i = idx / 4, e = idx % 4;
            packed[((size_t)i * NT + tid) * 4 + e] = W[(size_t)(p * KC + kk) * ncols + col0 + r * nh + u];
        }
    }
}

static float *s_up(csa_stoch *h, const float *src, size_t n, int &rc)
{
    void *p = nullptr;
    if (hipMalloc(&p, sizeof(float) * (n ? n : 1)) != hipSuccess) { rc = CSA_ERR_NOMEM; return nullptr; }
    h->owned.push_back(p);
    if (src && hipMemcpy(p, src, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
    return (float *)p;
}

// GRU5: weight_ih (nx,3H), weight_zh (H,3H), weight_encoder (H,2H), optional bias_ih (3H), bias_zh (3H)
extern "C" int csa_stoch_gru5_create(int nx, int nh, const float *weight_ih, const float *weight_zh,
                                     const float *weight_encoder, const float *bias_ih, const float *bias_zh,
                                     int max_rows, csa_stoch **out)
{
    if (!weight_ih || !weight_zh || !weight_encoder || !out || max_rows <= 0) { csa_set_error_msg("csa_stoch_gru5_create: bad argument"); return CSA_ERR_ARG; }
    if (!(nh == 64 || nh == 96 || nh == 128) || nx % 4) { csa_set_error_msg("csa_stoch_gru5_create: hidden size 64/96/128, nx multiple of 4"); return CSA_ERR_UNSUPPORTED; }
    csa_stoch *h = new csa_stoch();
    h->kind = 0; h->nx = nx; h->nh = nh; h->max_rows = max_rows;
    int rc = CSA_OK;
    std::vector<float> t((size_t)3 * nh * nx);
    for (int k = 0; k < nx; ++k) for (int n = 0; n < 3 * nh; ++n) t[(size_t)n * nx + k] = weight_ih[(size_t)k * 3 * nh + n];
    h->w_in_t = s_up(h, t.data(), t.size(), rc);
    h->b_in = bias_ih ? s_up(h, bias_ih, 3 * nh, rc) : nullptr;
    std::vector<float> pk((size_t)3 * nh * nh);
    stoch_pack_rows(nh, 2, weight_encoder, 2 * nh, 0, pk.data());
    h->wp_a = s_up(h, pk.data(), (size_t)2 * nh * nh, rc);
    stoch_pack_rows(nh, 3, weight_zh, 3 * nh, 0, pk.data());
    h->wp_b = s_up(h, pk.data(), (size_t)3 * nh * nh, rc);
    h->b_zh = bias_zh ? s_up(h, bias_zh, 3 * nh, rc) : nullptr;
    h->has_bias = (bias_ih && bias_zh) ? 1 : 0;
    h->host_a.assign(weight_zh, weight_zh + (size_t)nh * 3 * nh);            // kept for the BPTT packings (csa_stoch_enable_training)
    h->host_b.assign(weight_encoder, weight_encoder + (size_t)nh * 2 * nh);
    h->w_ref_in = s_up(h, weight_ih, (size_t)nx * 3 * nh, rc);
    h->XP = s_up(h, nullptr, (size_t)max_rows * 3 * nh, rc);
    if (rc) { for (void *p : h->owned) (void)hipFree(p); delete h; return rc; }
    *out = h;
    return CSA_OK;
}

// LSTM4: weight_encoder (nx + H, 5H)
extern "C" int csa_stoch_lstm4_create(int nx, int nh, const float *weight_encoder, int max_rows, csa_stoch **out)
{
    if (!weight_encoder || !out || max_rows <= 0) { csa_set_error_msg("csa_stoch_lstm4_create: bad argument"); return CSA_ERR_ARG; }
    if (!(nh == 64 || nh == 96 || nh == 128) || nx % 4) { csa_set_error_msg("csa_stoch_lstm4_create: hidden size 64/96/128, nx multiple of 4"); return CSA_ERR_UNSUPPORTED; }
    csa_stoch *h = new csa_stoch();
    h->kind = 1; h->nx = nx; h->nh = nh; h->max_rows = max_rows;
    int rc = CSA_OK;
    std::vector<float> t((size_t)5 * nh * nx);
    for (int k = 0; k < nx; ++k) for (int n = 0; n < 5 * nh; ++n) t[(size_t)n * nx + k] = weight_encoder[(size_t)k * 5 * nh + n];
    h->w_in_t = s_up(h, t.data(), t.size(), rc);
    h->b_in = nullptr;
    std::vector<float> pk((size_t)5 * nh * nh);
    stoch_pack_rows(nh, 5, weight_encoder + (size_t)nx * 5 * nh, 5 * nh, 0, pk.data());
    h->wp_a = s_up(h, pk.data(), pk.size(), rc);
    h->wp_b = nullptr; h->b_zh = nullptr;
    h->host_a.assign(weight_encoder + (size_t)nx * 5 * nh, weight_encoder + (size_t)(nx + nh) * 5 * nh);
    h->w_ref_in = s_up(h, weight_encoder, (size_t)nx * 5 * nh, rc);
    h->XP = s_up(h, nullptr, (size_t)max_rows * 5 * nh, rc);
    if (rc) { for (void *p : h->owned) (void)hipFree(p); delete h; return rc; }
    *out = h;
    return CSA_OK;
}

extern "C" int csa_stoch_destroy(csa_stoch *h)
{
    if (!h) return CSA_ERR_ARG;
    for (void *p : h->owned) (void)hipFree(p);
    delete h;
    return CSA_OK;
}

// x (T,B,nx), h0 (B,H), eps (T,B,H) -> out (T,B,H)
static int gru5_forward_impl(csa_stoch *h, int T, int B, const float *x, const float *h0, const float *eps, float *out, void *stream,
                             bool train)
{
    if (!h || h->kind != 0 || !x || !h0 || !eps || !out || T <= 0 || B <= 0 || (long)T * B > h->max_rows) { csa_set_error_msg("csa_stoch_gru5_forward: bad argument"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int nh = h->nh;
    int rc = launch_proj_gemm(x, h->w_in_t, h->b_in, h->XP, T * B, 3 * nh, h->nx, s);
    if (rc) return rc;
    const dim3 grid((B + 1) / 2), block(nh * 4);
    if (train && !h->Hseq) { csa_set_error_msg("csa_stoch_gru5_forward_train: call csa_stoch_enable_training first"); return CSA_ERR_ARG; }
#define G5(NHv, TR) hipLaunchKernelGGL((stoch_gru5_kernel<NHv, TR>), grid, block, 0, s, (const f32x4 *)h->wp_a, (const f32x4 *)h->wp_b, \
                                       h->b_zh, h->XP, eps, h0, out, B, T, h->ZN, h->Zs, h->EX, h->Hseq)
    switch (nh) {
    case 64:  if (train) G5(64, true); else G5(64, false); break;
    case 96:  if (train) G5(96, true); else G5(96, false); break;
    default:  if (train) G5(128, true); else G5(128, false); break;
    }
#undef G5
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
extern "C" int csa_stoch_gru5_forward(csa_stoch *h, int T, int B, const float *x, const float *h0, const float *eps,
                                      float *out, void *stream)
{
    return gru5_forward_impl(h, T, B, x, h0, eps, out, stream, false);
}
extern "C" int csa_stoch_gru5_forward_train(csa_stoch *h, int T, int B, const float *x, const float *h0, const float *eps,
                                            float *out, void *stream)
{
    return gru5_forward_impl(h, T, B, x, h0, eps, out, stream, true);
}

// x (T,B,nx), (h0,c0) (B,H), eps (T,B,H) -> out (T,B,H), hT, cT (B,H; nullable)
static int lstm4_forward_impl(csa_stoch *h, int T, int B, const float *x, const float *h0, const float *c0, const float *eps,
                              float *out, float *hT, float *cT, void *stream, bool train)
{
    if (!h || h->kind != 1 || !x || !h0 || !c0 || !eps || !out || T <= 0 || B <= 0 || (long)T * B > h->max_rows) { csa_set_error_msg("csa_stoch_lstm4_forward: bad argument"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int nh = h->nh;
    int rc = launch_proj_gemm(x, h->w_in_t, nullptr, h->XP, T * B, 5 * nh, h->nx, s);
    if (rc) return rc;
    const dim3 grid((B + 1) / 2), block(nh * 4);
    if (train && !h->Hseq) { csa_set_error_msg("csa_stoch_lstm4_forward_train: call csa_stoch_enable_training first"); return CSA_ERR_ARG; }
#define L4(NHv, TR) hipLaunchKernelGGL((stoch_lstm4_kernel<NHv, TR>), grid, block, 0, s, (const f32x4 *)h->wp_a, h->XP, eps, h0, c0, out, \
                                       hT, cT, B, T, h->Hseq, h->Cseq)
    switch (nh) {
    case 64:  if (train) L4(64, true); else L4(64, false); break;
    case 96:  if (train) L4(96, true); else L4(96, false); break;
    default:  if (train) L4(128, true); else L4(128, false); break;
    }
#undef L4
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
extern "C" int csa_stoch_lstm4_forward(csa_stoch *h, int T, int B, const float *x, const float *h0, const float *c0,
                                       const float *eps, float *out, float *hT, float *cT, void *stream)
{
    return lstm4_forward_impl(h, T, B, x, h0, c0, eps, out, hT, cT, stream, false);
}
extern "C" int csa_stoch_lstm4_forward_train(csa_stoch *h, int T, int B, const float *x, const float *h0, const float *c0,
                                             const float *eps, float *out, float *hT, float *cT, void *stream)
{
    return lstm4_forward_impl(h, T, B, x, h0, c0, eps, out, hT, cT, stream, true);
}
