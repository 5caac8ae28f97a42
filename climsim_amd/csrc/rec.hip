// rec.hip -- the level-recurrent half of nn.LSTM / nn.GRU (rnn/models/models.py:493,536):
// 60 dependent cell steps per column, gates = P[t] + W_hh h_{t-1}  (P = hoisted W_ih x_t + b).
//
// MI355X design: REGISTER-STATIONARY recurrent weights.  W_hh (4*128 x 128 fp32 = 256 KB) does
// not fit the 160 KB LDS but does fit the CU's 512 KB vector register file, so one workgroup of
// 4*nh threads (8 waves at nh=128, two per SIMD, <=256 VGPRs each) keeps the whole matrix in
// VGPRs for all 60 steps and never re-reads it.  Thread (u, p): hidden unit u = tid>>2, k-quarter
// p = tid&3; it holds the G gate rows of unit u restricted to k in [p*nh/4, (p+1)*nh/4) -- G*nh/4
// weights.  A workgroup advances TWO columns at once: h_{t-1} of both columns sits in LDS as
// (k, col) pairs, is read with ds_read_b128 (4 distinct addresses per wave -> broadcast), and one
// v_pk_fma_f32 (weight broadcast to both halves via op_sel) updates both columns' partial sums.
// The 4 k-quarters are then summed with two DPP quad-permute adds, lanes p<2 apply the gate
// non-linearities for column p, keep c_t in a register and publish h_t to the other LDS buffer:
// one workgroup barrier per step, no inter-workgroup communication, no atomics.
//
// Two columns per CU is the finest granularity at which the FP32 vector pipe is saturated
// (packed FMA) -- MFMA would need >=4 (4x4x1) or 16 (16x16x4) columns per CU and leaves most of
// the chip idle at the 384-column batch of BASELINE.json configs[1]; see DESIGN.md.
#include "common.h"

#ifndef CSA_FAST_GATES
#define CSA_FAST_GATES 1
#endif

__device__ __forceinline__ float sigmoid_f(float x)
{
#if CSA_FAST_GATES
    // v_exp_f32 (2^x, <=1 ulp) + v_rcp_f32 (1 ulp): absolute error ~1e-7, same size as the fp32
    // rounding of the O(1) gate values themselves.
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
#else
    return 1.0f / (1.0f + expf(-x));
#endif
}

__device__ __forceinline__ float tanh_f(float x)
{
#if CSA_FAST_GATES
    // tanh(x) = 1 - 2/(exp(2x)+1); for |x| < 0.04 the cancellation would cost relative accuracy,
    // so switch to the odd Taylor polynomial there (error < 1e-9 relative).
    const float e = __builtin_amdgcn_exp2f(2.88539008177792681f * x);
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    const float x2 = x * x;
    const float small = x * (1.0f + x2 * (-0.333333333f + x2 * 0.133333333f));
    return fabsf(x) < 0.04f ? small : big;
#else
    return tanhf(x);
#endif
}

__device__ __forceinline__ float quad_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    return v;
}

// acc(col0,col1) += w.x * h(col0,col1)   /   += w.y * h
#define PK_FMA_LO(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(h))
#define PK_FMA_HI(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(h))

// Packed weight layout (host packer below): thread tid owns G*KC weights, KC = NH/4, stored as
// float4 number i (0 <= i < G*KC/4) at Wp4[i*NT + tid]; within a thread the order is
// idx = g*KC + kk  <->  W_hh[g*NH + u][p*KC + kk].
template <int NH, int G>
__global__ __launch_bounds__(NH * 4, 2) void rec_kernel(
    const f32x4 *__restrict__ Wp4, const float *__restrict__ bhn, const float *__restrict__ P,
    const float *__restrict__ h0, const float *__restrict__ c0, float *__restrict__ Hout,
    int B, int L, int reverse_out)
{
    constexpr int NT = NH * 4;          // threads
    constexpr int KC = NH / 4;          // k per thread
    constexpr int CH = 2 * KC + 4;      // LDS floats per k-quarter (padded by one 16-B slot)
    static_assert(KC % 2 == 0, "nh must be a multiple of 8");
    __shared__ __attribute__((aligned(16))) float hbuf[2][4 * CH];

    const int tid = threadIdx.x, u = tid >> 2, p = tid & 3, col = p & 1;
    int b = 2 * blockIdx.x + col;
    const bool valid = b < B;
    if (!valid) b = B - 1;              // odd tail: the spare column recomputes a real one, writes nothing
    const bool writer = valid && p < 2;

    // ---- weights -> registers (read once) -------------------------------------------------
    f32x2 w[G][KC / 2];
#pragma unroll
    for (int i = 0; i < G * KC / 4; ++i) {
        const f32x4 v = Wp4[(size_t)i * NT + tid];
        const int g = (4 * i) / KC, kk = (4 * i) % KC;
        w[g][kk / 2] = f32x2{v.x, v.y};
        w[g][kk / 2 + 1] = f32x2{v.z, v.w};
    }
    float bn = 0.0f;
    if (G == 3) bn = bhn[u];

    // ---- initial state ---------------------------------------------------------------------
    float h = h0[(size_t)b * NH + u];
    float c = (G == 4) ? c0[(size_t)b * NH + u] : 0.0f;
    const int hslot = 2 * u + col + 4 * (u / KC);   // position of (k=u, col) in an hbuf
    if (p < 2) hbuf[0][hslot] = h;

    constexpr int PS = G == 4 ? 4 : 4;               // P row stride per unit (GRU rows padded to 4)
    const float *Pb = P + (size_t)b * (PS * NH) + u * PS;
    const size_t Pstep = (size_t)B * (PS * NH);
    f32x4 pre = *(const f32x4 *)Pb;
    __syncthreads();

    for (int t = 0; t < L; ++t) {
        const int cur = t & 1;
        // prefetch next step's input projection (independent of the recurrence)
        f32x4 pre_next = pre;
        if (t + 1 < L) pre_next = *(const f32x4 *)(Pb + (size_t)(t + 1) * Pstep);

        f32x2 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = f32x2{0.0f, 0.0f};
        const f32x4 *hp = (const f32x4 *)&hbuf[cur][p * CH];
#pragma unroll
        for (int j = 0; j < KC / 2; ++j) {
            const f32x4 hv = hp[j];
            const f32x2 ha = {hv.x, hv.y}, hb = {hv.z, hv.w};   // (k,col0|col1), (k+1,col0|col1)
#pragma unroll
            for (int g = 0; g < G; ++g) PK_FMA_LO(acc[g], w[g][j], ha);
#pragma unroll
            for (int g = 0; g < G; ++g) PK_FMA_HI(acc[g], w[g][j], hb);
        }
        float s[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float sx = quad_sum(acc[g].x), sy = quad_sum(acc[g].y);
            s[g] = col ? sy : sx;
        }
        if (G == 4) {
            const float ig = sigmoid_f(pre.x + s[0]);
            const float fg = sigmoid_f(pre.y + s[1]);
            const float gg = tanh_f(pre.z + s[2]);
            const float og = sigmoid_f(pre.w + s[G - 1]);
            c = fg * c + ig * gg;
            h = og * tanh_f(c);
        } else {
            const float r = sigmoid_f(pre.x + s[0]);
            const float z = sigmoid_f(pre.y + s[1]);
            const float n = tanh_f(pre.z + r * (s[2] + bn));
            h = (1.0f - z) * n + z * h;
        }
        if (p < 2) hbuf[cur ^ 1][hslot] = h;
        if (writer) {
            const int lvl = reverse_out ? L - 1 - t : t;
            Hout[((size_t)lvl * B + b) * NH + u] = h;
        }
        pre = pre_next;
        __syncthreads();
    }
}

size_t rec_packed_floats(int use_lstm, int nh) { return (size_t)(use_lstm ? 4 : 3) * nh * nh; }

void rec_pack_weights(int use_lstm, int nh, const float *w_hh, float *packed)
{
    const int G = use_lstm ? 4 : 3, NT = nh * 4, KC = nh / 4;
    for (int tid = 0; tid < NT; ++tid) {
        const int u = tid >> 2, p = tid & 3;
        for (int idx = 0; idx < G * KC; ++idx) {
            const int g = idx / KC, kk = idx % KC;
            const int i = idx / 4, e = idx % 4;
            packed[((size_t)i * NT + tid) * 4 + e] = w_hh[(size_t)(g * nh + u) * nh + p * KC + kk];
        }
    }
}

template <int NH>
static int launch_rec_nh(int use_lstm, const float *whh, const float *bhn, const float *P, const float *h0,
                         const float *c0, float *Hout, int B, int L, int reverse_out, hipStream_t s)
{
    const dim3 grid((B + 1) / 2), block(NH * 4);
    if (use_lstm)
        hipLaunchKernelGGL((rec_kernel<NH, 4>), grid, block, 0, s, (const f32x4 *)whh, bhn, P, h0, c0, Hout, B, L,
                           reverse_out);
    else
        hipLaunchKernelGGL((rec_kernel<NH, 3>), grid, block, 0, s, (const f32x4 *)whh, bhn, P, h0, c0, Hout, B, L,
                           reverse_out);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

int launch_rec(int use_lstm, int nh, const float *whh_packed, const float *bhn, const float *P,
               const float *h0, const float *c0, float *Hout, int B, int L, int reverse_out, hipStream_t s)
{
    switch (nh) {
    case 64:  return launch_rec_nh<64>(use_lstm, whh_packed, bhn, P, h0, c0, Hout, B, L, reverse_out, s);
    case 96:  return launch_rec_nh<96>(use_lstm, whh_packed, bhn, P, h0, c0, Hout, B, L, reverse_out, s);
    case 128: return launch_rec_nh<128>(use_lstm, whh_packed, bhn, P, h0, c0, Hout, B, L, reverse_out, s);
    default:
        csa_set_error_msg("rec: hidden size not supported by the register-stationary kernel (64, 96, 128)");
        return CSA_ERR_UNSUPPORTED;
    }
}
