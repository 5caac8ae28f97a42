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
#include <type_traits>
#include <cstring>

#ifndef CSA_FAST_GATES
#define CSA_FAST_GATES 1
#endif

// The activation primitives of the second-generation kernels: 2^x and 1/x.  CSA_FAST_GATES=1 (default): v_exp_f32 and
// v_rcp_f32 (1 ulp each); CSA_FAST_GATES=0 (diagnostic build, tools/build_exact_gates.sh): libm exp2f and IEEE division --
// used once per round to show that the gate approximations are not what separates the HIP path from the reference
// (profiles/r2_stagewise_parity.txt).
#if CSA_FAST_GATES
#define CSA_EXP2(x) __builtin_amdgcn_exp2f(x)
#define CSA_RCP(x) __builtin_amdgcn_rcpf(x)
#else
#define CSA_EXP2(x) exp2f(x)
#define CSA_RCP(x) (1.0f / (x))
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
    // tanh(x) = (1 - t)/(1 + t), t = exp(-2x): absolute error ~1 ulp of 1 for every x.
    const float t = fminf(__builtin_amdgcn_exp2f(-2.88539008177792681f * x), 1e30f);
    return (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
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

__device__ __forceinline__ float dpp_xor1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

// Workgroup barrier that waits for this wave's LDS traffic ONLY.  __syncthreads() also drains
// vmcnt(0), i.e. it would stall every step on the global prefetch of P and on the h_t store
// (measured: +475 ns per step); those stay in flight across the barrier here.
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// acc(col0,col1) += w.x * h(col0,col1)   /   += w.y * h
#define PK_FMA_LO(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(h))
#define PK_FMA_HI(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(h))

// Packed weight layout (host packer below): thread tid owns G*KC weights, KC = NH/4, stored as
// float4 number i (0 <= i < G*KC/4) at Wp4[i*NT + tid]; within a thread the order is
// idx = g*KC + kk  <->  W_hh[g*NH + u][p*KC + kk].
// TRAIN (GRU): additionally saves what BPTT needs -- [r, z, n, W_hn h + b_hn] written IN PLACE over the (4-padded)
// pre-activation row P(t,b,u,:) it came from, and h_t into Hseq (L+1 slots in SEQUENCE order, slot 0 = h_init).
template <int NH, int G, bool TRAIN = false>
__global__ __launch_bounds__(NH * 4, (NH / 16 + 3) / 4) void rec_kernel(
    const f32x4 *__restrict__ Wp4, const float *__restrict__ bhn, const float *P,
    const float *__restrict__ h0, const float *__restrict__ c0, float *__restrict__ Hout,
    int B, int L, int reverse_out, float *Pw = nullptr, float *__restrict__ Hseq = nullptr)
{
    constexpr int NT = NH * 4;          // threads
    constexpr int KC = NH / 4;          // k per thread
    constexpr int CH = 2 * KC + 4;      // LDS floats per k-quarter (padded by one 16-B slot)
    static_assert(KC % 2 == 0, "nh must be a multiple of 8");
    __shared__ __attribute__((aligned(16))) float hbuf[2][4 * CH];

    const int tid = threadIdx.x, u = tid >> 2, p = tid & 3, col = p & 1;
#ifdef REC_EXP_CLOCK
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = wall_clock64();
#endif
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
    if (TRAIN && writer) Hseq[(size_t)b * NH + u] = h;

    constexpr int PS = G == 4 ? 4 : 4;               // P row stride per unit (GRU rows padded to 4)
    const float *Pb = P + (size_t)b * (PS * NH) + u * PS;
    const size_t Pstep = (size_t)B * (PS * NH);
    // input projections are prefetched two steps ahead (they do not depend on the recurrence)
    f32x4 pre = *(const f32x4 *)Pb;
    f32x4 pre1 = L > 1 ? *(const f32x4 *)(Pb + Pstep) : pre;
    __syncthreads();

    for (int t = 0; t < L; ++t) {
        const int cur = t & 1;
        // h_{t-1} is stored one step late, just ahead of the prefetch: by the time the register
        // rotation at the bottom of the loop needs this step's prefetch (an in-order vmcnt wait)
        // both have long completed, so no step ever stalls on global memory.
#ifndef REC_EXP_NO_GLOBAL
        if (t > 0 && writer) {
            const int lvl = reverse_out ? L - t : t - 1;
            Hout[((size_t)lvl * B + b) * NH + u] = h;
            if (TRAIN) Hseq[((size_t)t * B + b) * NH + u] = h;
        }
#endif
        f32x4 pre2 = pre1;
#ifndef REC_EXP_NO_GLOBAL
        if (t + 2 < L) pre2 = *(const f32x4 *)(Pb + (size_t)(t + 2) * Pstep);
#endif

        f32x2 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = f32x2{0.0f, 0.0f};
        const f32x4 *hp = (const f32x4 *)&hbuf[cur][p * CH];
#ifdef REC_EXP_NO_FMA
#pragma unroll
        for (int j = 0; j < 1; ++j) {
#else
#pragma unroll
        for (int j = 0; j < KC / 2; ++j) {
#endif
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
#ifdef REC_EXP_NO_GATES
        h = 0.25f * (pre.x + s[0]) + 0.01f * (pre.y + s[1] + pre.z + s[2] + pre.w + s[G - 1]);
        h = fminf(fmaxf(h, -1.0f), 1.0f);
#else
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
            const float hn = s[2] + bn;
            const float n = tanh_f(pre.z + r * hn);
            h = (1.0f - z) * n + z * h;
            if (TRAIN && writer) *(f32x4 *)(Pw + ((size_t)t * B + b) * (PS * NH) + u * PS) = f32x4{r, z, n, hn};
        }
#endif
        if (p < 2) hbuf[cur ^ 1][hslot] = h;
        pre = pre1;
        pre1 = pre2;
        LDS_BARRIER();
    }
    if (writer) {
        const int lvl = reverse_out ? 0 : L - 1;
        Hout[((size_t)lvl * B + b) * NH + u] = h;
        if (TRAIN) Hseq[((size_t)L * B + b) * NH + u] = h;
    }
#ifdef REC_EXP_CLOCK
    if (tid == 0 && blockIdx.x == 0) {   // diagnostic build only: shader cycles and 100 MHz ticks of block 0
        Hout[0] = (float)(__builtin_readcyclecounter() - clk0);
        Hout[1] = (float)(wall_clock64() - rt0);
    }
#endif
}


// ------------------------------------------------------------------------------------------------
// LSTM kernel, second generation: same register-stationary / two-columns-per-CU layout, with the
// serial part of every step (everything that is not the 128 packed FMAs) cut down:
//   * h_{t-1} is kept in LDS twice, as (col0,col1) and as (col1,col0) pairs; odd k-quarters read
//     the swapped copy, so in every lane accumulator.x is "my column" (col = p&1).  The k-quarter
//     sum then becomes a reduce-SCATTER of 6 DPP adds (was an all-reduce of 16): after
//     r = acc.x + xor1(acc.y) and f = r[s] + xor2(r[s+2]) lane p holds the two gates of ITS group
//     (p<2: i,g~; p>=2: f,o) for ITS column.  Which gate sits in which accumulator slot differs per
//     lane group and is decided by the host packer (slots = [i,g,f,o] for p<2, [f,o,i,g] for p>=2),
//     so no select is needed anywhere.
//   * two activations per lane instead of five (slot 0 is always a sigmoid, slot 1 is tanh or
//     sigmoid through per-lane constants), one DPP transfer of i*g~ to the lane that owns c_t.
//   * no register rotation of the prefetched projections (time loop unrolled x2), so the compiler
//     never waits on a just-issued global load or store inside the recurrence.
// P rows are laid out [i, g~, f, o] per unit (see pack_ih in api.hip): lane reads one 8-byte pair.
// TRAIN = true additionally saves what BPTT needs: the activated gates are written IN PLACE over
// the pre-activations P(t,b,u,[i,g~,f,o]) they came from, c_t goes to Cseq (L+1 slots, slot 0 = c_init),
// h_t also to Hseq (L+1 slots in SEQUENCE order, slot 0 = h_init; may alias Hout for the downward RNN).
// NL4 > 0 (nh = 144, the reference's default config): 9 waves per workgroup put three waves on one SIMD, so
// a wave may hold 168 VGPRs, not 256, and the 144 weights per lane no longer fit beside the working set.  The
// last NL4 float4 of every slot's weight run then live in LDS instead (NL4*64 B per lane, read conflict-free
// once per step); the other (KC/4 - NL4) float4 per slot stay register-stationary.
template <int NH, bool TRAIN, int NL4 = 0>
__global__ __launch_bounds__(NH * 4, (NH / 16 + 3) / 4) void lstm_rec2_kernel(
    const f32x4 *__restrict__ Wp4, float *__restrict__ P,
    const float *__restrict__ h0, const float *__restrict__ c0, float *__restrict__ Hout,
    int B, int L, int reverse_out, float *__restrict__ Hseq, float *__restrict__ Cseq)
{
    constexpr int NT = NH * 4;
    constexpr int KC = NH / 4;
    constexpr int CH = 2 * KC + 4;      // floats per k-quarter, padded by one 16-B slot
    constexpr int CPY = 4 * CH;         // floats per copy
    static_assert(KC % 4 == 0, "nh must be a multiple of 16");
    constexpr int KR = KC / 2 - 2 * NL4; // weight pairs per slot kept in registers
    __shared__ __attribute__((aligned(16))) float hbuf[2][2 * CPY];
    extern __shared__ f32x4 wlds[];      // NL4 > 0: 4*NL4*NT float4 of dynamic LDS

    const int tid = threadIdx.x, u = tid >> 2, p = tid & 3, col = p & 1, grp = p >> 1;
#ifdef REC_EXP_CLOCK
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = wall_clock64();
#endif
    int b = 2 * blockIdx.x + col;
    const bool valid = b < B;
    if (!valid) b = B - 1;
    const bool owner = grp == 1;                 // lanes p>=2 own c_t / h_t of (u, col)

    f32x2 w[4][KR > 0 ? KR : 1];
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        const f32x4 v = Wp4[(size_t)i * NT + tid];
        const int s = (4 * i) / KC, kk = (4 * i) % KC;
        if (kk / 2 < KR) {
            w[s][kk / 2] = f32x2{v.x, v.y};
            w[s][kk / 2 + 1] = f32x2{v.z, v.w};
        } else {
            wlds[(s * NL4 + (kk / 2 - KR) / 2) * NT + tid] = v;
        }
    }
    // slot-1 activation: tanh for the (i,g~) lanes, sigmoid for the (f,o) lanes, both written as
    // (1 - nb*t) / (1 + t) with t = exp(-x) (sigmoid, nb = 0) or t = exp(-2x) (tanh, nb = 1).  The
    // tanh numerator 1 - t keeps an absolute error of one ulp of 1 (6e-8) for every x, unlike
    // 2*sigmoid(2x) - 1; t is clamped so that exp overflow gives -1, not NaN.
    const float k1 = grp ? -1.44269504088896341f : -2.88539008177792681f;
    const float nb1 = grp ? 0.0f : 1.0f;

    float h = h0[(size_t)b * NH + u];
    float c = c0[(size_t)b * NH + u];
    const int slotN = 2 * u + col + 4 * (u / KC);
    const int slotS = CPY + 2 * u + (1 - col) + 4 * (u / KC);
    if (owner) { hbuf[0][slotN] = h; hbuf[0][slotS] = h; }
    if (TRAIN && owner && valid) {
        Hseq[(size_t)b * NH + u] = h;
        Cseq[(size_t)b * NH + u] = c;
    }

    const float *Pb = P + (size_t)b * (4 * NH) + u * 4 + grp * 2;
    const size_t Pstep = (size_t)B * (4 * NH);
    const int rdoff = col * CPY + p * CH;
    f32x2 preA = *(const f32x2 *)Pb, preB = preA;
    __syncthreads();

    // P(t+1) is fetched with an asm global_load at the top of step t and first touched behind an
    // explicit vmcnt(1) ("everything but the newest VMEM op") just before the gates of step t+1,
    // a whole step later; the h_t store of the previous step is older still.  hipcc inserts no
    // waits for loads issued inside asm, so nothing in the recurrence ever stalls on global
    // memory; the "+v" operand of the wait orders the consumers behind it.  The prefetch is
    // UNCONDITIONAL (the last step re-reads its own row) and so is the wait: every step has the
    // same loads in flight and ONE wait statement with ONE count, so the compiler never sees a
    // path on which a prefetch register is live without its wait (the hazard recorded at
    // lstm_rec4_kernel; tools/check_asm_prefetch.py checks the emitted ISA of every such kernel).  The LDS reads of
    // h_{t-1} are left to the compiler (two ds_read_b128 in flight per wave).  Measured with the
    // in-kernel stamps of the REC_EXP_STAMP build: hand-pipelining them 4 deep changes nothing and
    // issuing all sixteen up front costs +170 cycles per step; the FMA phase is bound by the
    // LDS->VGPR return traffic (16 KB per wave per step) competing with the packed FMAs, not by
    // read latency: 1234 cycles per step with h in registers vs ~1650 with h from LDS
    // (tools/valu_bench.hip).
#ifdef REC_EXP_HALF_LDS   /* diagnostic: half the LDS reads, same FMA count (results are wrong) */
#define REC_HIDX(j) ((j) & ~1)
#elif defined(REC_EXP_QUARTER_LDS)
#define REC_HIDX(j) ((j) & ~3)
#else
#define REC_HIDX(j) (j)
#endif
#define LSTM2_STEP(T, CUR, NXT)                                                                    \
    {                                                                                              \
        const int t_ = (T);                                                                        \
        {                                                                                          \
            const float *pn = Pb + (size_t)(t_ + 1 < L ? t_ + 1 : L - 1) * Pstep;                  \
            asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(NXT) : "v"(pn) : "memory");     \
        }                                                                                          \
        const f32x4 *hp = (const f32x4 *)&hbuf[t_ & 1][rdoff];                                     \
        f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};                           \
        _Pragma("unroll") for (int j = 0; j < KR; ++j) {                                           \
            const f32x4 hv = hp[REC_HIDX(j)];                                                      \
            const f32x2 ha = {hv.x, hv.y}, hb = {hv.z, hv.w};                                      \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) PK_FMA_LO(acc[s], w[s][j], ha);          \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) PK_FMA_HI(acc[s], w[s][j], hb);          \
        }                                                                                          \
        _Pragma("unroll") for (int q = 0; q < NL4; ++q) {                                          \
            const f32x4 h0v = hp[KR + 2 * q], h1v = hp[KR + 2 * q + 1];                            \
            const f32x2 ha0 = {h0v.x, h0v.y}, hb0 = {h0v.z, h0v.w};                                \
            const f32x2 ha1 = {h1v.x, h1v.y}, hb1 = {h1v.z, h1v.w};                                \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                        \
                const f32x4 wv = wlds[(s * NL4 + q) * NT + tid];                                   \
                const f32x2 w0 = {wv.x, wv.y}, w1 = {wv.z, wv.w};                                  \
                PK_FMA_LO(acc[s], w0, ha0);                                                        \
                PK_FMA_HI(acc[s], w0, hb0);                                                        \
                PK_FMA_LO(acc[s], w1, ha1);                                                        \
                PK_FMA_HI(acc[s], w1, hb1);                                                        \
            }                                                                                      \
        }                                                                                          \
        STAMP(1, acc[0])                                                                           \
        float r[4];                                                                                \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) r[s] = acc[s].x + dpp_xor1(acc[s].y);        \
        asm volatile("s_waitcnt vmcnt(1)" : "+v"(CUR));                                            \
        const float v0 = r[0] + dpp_xor2(r[2]) + CUR.x;                                            \
        const float v1 = r[1] + dpp_xor2(r[3]) + CUR.y;                                            \
        const float g0 = CSA_RCP(1.0f + CSA_EXP2(-1.44269504088896341f * v0)); \
        const float t1 = fminf(CSA_EXP2(k1 * v1), 1e30f);                            \
        const float g1 = (1.0f - nb1 * t1) * CSA_RCP(1.0f + t1);                     \
        const float ig = dpp_xor2(g0 * g1);           /* sigma(i)*tanh(g~) arrives at the (f,o) lane */ \
        c = g0 * c + ig;                                                                           \
        const float tc = fminf(CSA_EXP2(-2.88539008177792681f * c), 1e30f);          \
        const float th = (1.0f - tc) * CSA_RCP(1.0f + tc);                           \
        h = g1 * th;                                                                               \
        STAMP(2, h)                                                                                \
        if (owner) {                                                                               \
            hbuf[(t_ & 1) ^ 1][slotN] = h;                                                         \
            hbuf[(t_ & 1) ^ 1][slotS] = h;                                                         \
            if (valid) Hout[((size_t)(reverse_out ? L - 1 - t_ : t_) * B + b) * NH + u] = h;       \
            if (TRAIN && valid) {                                                                  \
                Hseq[((size_t)(t_ + 1) * B + b) * NH + u] = h;                                     \
                Cseq[((size_t)(t_ + 1) * B + b) * NH + u] = c;                                     \
            }                                                                                      \
        }                                                                                          \
        if (TRAIN && valid)                                                                        \
            *(f32x2 *)(P + ((size_t)t_ * B + b) * (4 * NH) + u * 4 + grp * 2) = f32x2{g0, g1};     \
        LDS_BARRIER();                                                                             \
        STAMP(0, h)                                                                                \
    }

#ifdef REC_EXP_STAMP
    // diagnostic build only: per-phase shader-cycle sums of one wave (phase k ends at stamp k)
    unsigned long long st_last = __builtin_readcyclecounter(), st_sum[3] = {0, 0, 0};
#define STAMP(K, DEP)                                                                              \
    {                                                                                              \
        asm volatile("" : "+v"(DEP));                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        const unsigned long long now_ = __builtin_readcyclecounter();                              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        st_sum[K] += now_ - st_last;                                                               \
        st_last = now_;                                                                            \
    }
#else
#define STAMP(K, DEP)
#endif

    for (int t = 0; t < L; t += 2) {
        LSTM2_STEP(t, preA, preB)
        if (t + 1 < L) LSTM2_STEP(t + 1, preB, preA)
    }
#undef LSTM2_STEP
#undef STAMP
#ifdef REC_EXP_CLOCK
    if (tid == 0 && blockIdx.x == 0) {
        Hout[0] = (float)(__builtin_readcyclecounter() - clk0);
        Hout[1] = (float)(wall_clock64() - rt0);
    }
#endif
#ifdef REC_EXP_STAMP
    if ((tid & 63) == 0 && blockIdx.x == 0) {
        float *dbg = Hout + 8 + (tid >> 6) * 4;
        dbg[0] = (float)st_sum[1]; dbg[1] = (float)st_sum[2]; dbg[2] = (float)st_sum[0];
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// LSTM kernel for LARGE batches (inference): FOUR columns per workgroup = two column pairs that share the
// register-stationary weights.  Per step the 128 packed FMAs are issued twice (once per pair, same weights, two LDS
// reads per k-step instead of one), but the barrier, the reduce-scatter and the activation chains of the two pairs are
// independent and interleave, so the ~900 non-FMA cycles of a lstm_rec2_kernel step are paid once per four columns
// instead of once per two.  Every column sees exactly the arithmetic of lstm_rec2_kernel (same lane -> weight mapping,
// same summation order), so results are bit-identical to it.  Used from 1,024 columns per call, where workgroups
// outnumber the CUs anyway.
template <int NH>
__global__ __launch_bounds__(NH * 4, (NH / 16 + 3) / 4) void lstm_rec4_kernel(
    const f32x4 *__restrict__ Wp4, const float *__restrict__ P,
    const float *__restrict__ h0, const float *__restrict__ c0, float *__restrict__ Hout,
    int B, int L, int reverse_out)
{
    constexpr int NT = NH * 4;
    constexpr int KC = NH / 4;
    constexpr int CH = 2 * KC + 4;
    constexpr int CPY = 4 * CH;
    constexpr int KR = KC / 2;
    static_assert(KC % 4 == 0, "nh must be a multiple of 16");
    __shared__ __attribute__((aligned(16))) float hbuf[2][2][2 * CPY];     // [parity][pair][two copies]

    const int tid = threadIdx.x, u = tid >> 2, p = tid & 3, col = p & 1, grp = p >> 1;
    int bA = 4 * blockIdx.x + col, bB = bA + 2;
    const bool validA = bA < B, validB = bB < B;
    if (!validA) bA = B - 1;
    if (!validB) bB = B - 1;
    const bool owner = grp == 1;

    f32x2 w[4][KR];
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        const f32x4 v = Wp4[(size_t)i * NT + tid];
        const int s = (4 * i) / KC, kk = (4 * i) % KC;
        w[s][kk / 2] = f32x2{v.x, v.y};
        w[s][kk / 2 + 1] = f32x2{v.z, v.w};
    }
    const float k1 = grp ? -1.44269504088896341f : -2.88539008177792681f;
    const float nb1 = grp ? 0.0f : 1.0f;

    float hA = h0[(size_t)bA * NH + u], cA = c0[(size_t)bA * NH + u];
    float hB = h0[(size_t)bB * NH + u], cB = c0[(size_t)bB * NH + u];
    const int slotN = 2 * u + col + 4 * (u / KC);
    const int slotS = CPY + 2 * u + (1 - col) + 4 * (u / KC);
    if (owner) {
        hbuf[0][0][slotN] = hA; hbuf[0][0][slotS] = hA;
        hbuf[0][1][slotN] = hB; hbuf[0][1][slotS] = hB;
    }
    const float *PbA = P + (size_t)bA * (4 * NH) + u * 4 + grp * 2;
    const float *PbB = P + (size_t)bB * (4 * NH) + u * 4 + grp * 2;
    const size_t Pstep = (size_t)B * (4 * NH);
    const int rdoff = col * CPY + p * CH;
    f32x2 preA0 = *(const f32x2 *)PbA, preA1 = preA0, preB0 = *(const f32x2 *)PbB, preB1 = preB0;
    __syncthreads();

#define LSTM4_GATES(R, CUR, C, H)                                                                  \
    {                                                                                              \
        const float v0 = R[0] + dpp_xor2(R[2]) + CUR.x;                                            \
        const float v1 = R[1] + dpp_xor2(R[3]) + CUR.y;                                            \
        const float g0 = CSA_RCP(1.0f + CSA_EXP2(-1.44269504088896341f * v0)); \
        const float t1 = fminf(CSA_EXP2(k1 * v1), 1e30f);                            \
        const float g1 = (1.0f - nb1 * t1) * CSA_RCP(1.0f + t1);                     \
        const float ig = dpp_xor2(g0 * g1);                                                        \
        C = g0 * C + ig;                                                                           \
        const float tc = fminf(CSA_EXP2(-2.88539008177792681f * C), 1e30f);          \
        const float th = (1.0f - tc) * CSA_RCP(1.0f + tc);                           \
        H = g1 * th;                                                                               \
    }
#define LSTM4_STEP(T, CURA, NXTA, CURB, NXTB)                                                      \
    {                                                                                              \
        const int t_ = (T);                                                                        \
        {   /* UNCONDITIONAL prefetch (the last step re-reads its own row): the number of loads in flight is then the  \
               same at every step and ONE wait statement serves all of them.  With a second, conditional wait        \
               ("vmcnt(0)" on the last step) hipcc placed register copies of the in-flight values BEFORE that wait:  \
               stale projections at t = L-1 whenever the load was late (seen only under memory contention). */      \
            const int tn_ = t_ + 1 < L ? t_ + 1 : L - 1;                                           \
            const float *pa = PbA + (size_t)tn_ * Pstep, *pb = PbB + (size_t)tn_ * Pstep;          \
            asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(NXTA) : "v"(pa) : "memory");    \
            asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(NXTB) : "v"(pb) : "memory");    \
        }                                                                                          \
        const f32x4 *hpA = (const f32x4 *)&hbuf[t_ & 1][0][rdoff];                                 \
        const f32x4 *hpB = (const f32x4 *)&hbuf[t_ & 1][1][rdoff];                                 \
        f32x2 accA[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};                          \
        f32x2 accB[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};                          \
        _Pragma("unroll") for (int j = 0; j < KR; ++j) {                                           \
            const f32x4 hvA = hpA[j], hvB = hpB[j];                                                \
            const f32x2 haA = {hvA.x, hvA.y}, hbA = {hvA.z, hvA.w};                                \
            const f32x2 haB = {hvB.x, hvB.y}, hbB = {hvB.z, hvB.w};                                \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) PK_FMA_LO(accA[s], w[s][j], haA);        \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) PK_FMA_LO(accB[s], w[s][j], haB);        \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) PK_FMA_HI(accA[s], w[s][j], hbA);        \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) PK_FMA_HI(accB[s], w[s][j], hbB);        \
        }                                                                                          \
        float rA[4], rB[4];                                                                        \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                            \
            rA[s] = accA[s].x + dpp_xor1(accA[s].y);                                               \
            rB[s] = accB[s].x + dpp_xor1(accB[s].y);                                               \
        }                                                                                          \
        asm volatile("s_waitcnt vmcnt(2)" : "+v"(CURA), "+v"(CURB));                               \
        LSTM4_GATES(rA, CURA, cA, hA)                                                              \
        LSTM4_GATES(rB, CURB, cB, hB)                                                              \
        if (owner) {                                                                               \
            hbuf[(t_ & 1) ^ 1][0][slotN] = hA;                                                     \
            hbuf[(t_ & 1) ^ 1][0][slotS] = hA;                                                     \
            hbuf[(t_ & 1) ^ 1][1][slotN] = hB;                                                     \
            hbuf[(t_ & 1) ^ 1][1][slotS] = hB;                                                     \
            const size_t lv = (size_t)(reverse_out ? L - 1 - t_ : t_) * B;                         \
            if (validA) Hout[(lv + bA) * NH + u] = hA;                                             \
            if (validB) Hout[(lv + bB) * NH + u] = hB;                                             \
        }                                                                                          \
        LDS_BARRIER();                                                                             \
    }
    for (int t = 0; t < L; t += 2) {
        LSTM4_STEP(t, preA0, preA1, preB0, preB1)
        if (t + 1 < L) LSTM4_STEP(t + 1, preA1, preA0, preB1, preB0)
    }
#undef LSTM4_STEP
#undef LSTM4_GATES
}

// ------------------------------------------------------------------------------------------------
// LSTM kernel for LARGE batches on the MATRIX pipe: four columns per workgroup through v_mfma_f32_4x4x1_16b_f32.
// That instruction multiplies 16 independent (4x1)·(1x4) blocks per issue: block = hidden unit, the 4 rows of a block = the
// unit's gate rows (i, f, g, o) at one k, the 4 columns = the workgroup's 4 grid columns.  So with
//     lane = 4*unit_in_wave + x:   A operand  W_hh[gate x of the unit][k]      (128 registers per lane, stationary)
//                                  B operand  h_{t-1}[k] of column x           (same for all 16 blocks)
//                                  D (4 regs) the FOUR gate sums of (unit, column x)
// a step is 128 MFMAs per wave (8 cycles each: the full fp32 rate, 64 FLOP/clk/SIMD, at FOUR columns per CU -- the packed-FMA
// kernels need the VALU for that), the k-sum needs no cross-lane reduction at all and every lane ends the step holding all
// four gate pre-activations of ITS (unit, column): no DPP exchange, c_t and h_t stay in that lane.
// B operands come from LDS; with BLGP one ds_read_b128 feeds SIXTEEN MFMAs: each 16-lane group of the wave reads a different
// k-quad of its column, and `blgp:4+g` makes the matrix pipe take B from lane group g for all four groups (16 KB of LDS
// return traffic per wave and step instead of 64 KB).
// Same arithmetic as the other LSTM kernels up to the order of the k-sum (four interleaved accumulator chains).
// TRAIN (round 3): the training forward at shard size -- the lane already ends a step with all four activated gates of its (unit,
// column): they are written IN PLACE over the projection row [i, g~, f, o], c_t to Cseq and h_t to Hseq (L+1 slots, slot 0 = the
// initial state), exactly what lstm_rec2_kernel<NH, true> saves and lstm_bwd_rec_kernel reads.
// NWL > 0 (nh = 144, the reference's default width): 9 waves put three on one SIMD, so a wave may hold 168 VGPRs and 144 weights +
// the working set do not fit; the weights of the LAST NWL k-values live in dynamic LDS (NWL/4 float4 per lane, consecutive lanes
// 16 bytes apart: conflict-free) and pass through four temporaries per 16-k block.
#ifndef REC4M_NWL144
#define REC4M_NWL144 32      /* k-values of the nh = 144 weight run kept in LDS */
#endif
template <int NH, bool BLGP, bool TRAIN = false, int NWL = 0>
__global__ __launch_bounds__(NH * 4, (NH / 16 + 3) / 4) void lstm_rec4m_kernel(
    const float *__restrict__ Wk, float *P,
    const float *__restrict__ h0, const float *__restrict__ c0, float *__restrict__ Hout,
    int B, int L, int reverse_out, float *__restrict__ Hseq = nullptr, float *__restrict__ Cseq = nullptr)
{
    constexpr int NT = NH * 4, NR = NH - NWL;
    static_assert(NH % 16 == 0 && NWL % 16 == 0, "nh and the LDS-resident tail must be multiples of 16");
    // h_{t-1} in LDS as [k / 16][(k % 16) / 4][column][k % 4]: the sixteen 16-byte chunks one ds_read_b128 of a wave touches
    // (4 lane groups x 4 columns) are 256 CONTIGUOUS bytes -> conflict-free however the hardware splits the wave
    // (a [column][k] layout with padded rows measured 50 % SQ_LDS_BANK_CONFLICT, profiles/r2_v4_memory_2700_sq_pmc.json)
    __shared__ __attribute__((aligned(16))) float hbuf[2][4 * NH];
    extern __shared__ f32x4 wl4[];        // NWL > 0: [NWL / 4][NT] float4
    static_assert(NWL == 0 || BLGP, "the LDS weight tail is written for the blgp operand-broadcast variant");

    const int tid = threadIdx.x, u = tid >> 2, x = tid & 3, lane = tid & 63;
    int b = 4 * blockIdx.x + x;
    const bool valid = b < B;
    if (!valid) b = B - 1;

    float w[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) w[k] = Wk[(size_t)k * NT + tid];
#pragma unroll
    for (int q = 0; q < NWL / 4; ++q)
        wl4[q * NT + tid] = f32x4{Wk[(size_t)(NR + 4 * q) * NT + tid], Wk[(size_t)(NR + 4 * q + 1) * NT + tid],
                                  Wk[(size_t)(NR + 4 * q + 2) * NT + tid], Wk[(size_t)(NR + 4 * q + 3) * NT + tid]};

    float h = h0[(size_t)b * NH + u], c = c0[(size_t)b * NH + u];
    const int hslot = (u >> 4) * 64 + ((((u & 15) >> 2) * 4 + x) << 2) + (u & 3);      // where (k = u, column x) lives
    hbuf[0][hslot] = h;
    if (TRAIN && valid) { Hseq[(size_t)b * NH + u] = h; Cseq[(size_t)b * NH + u] = c; }
    const float *Pb = P + (size_t)b * (4 * NH) + u * 4;
    const size_t Pstep = (size_t)B * (4 * NH);
    f32x4 preA = *(const f32x4 *)Pb, preB = preA;
    const int hoff = BLGP ? (((lane >> 4) * 4 + x) << 2) : (x << 2);                       // this lane's chunk of a 16-k block
    __syncthreads();

#ifndef REC4M_EXP_MFMA_DIV      /* diagnostic builds (tools/rec_bench): 2 = half of the MFMAs, results are wrong */
#define REC4M_EXP_MFMA_DIV 1
#endif
#ifdef REC4M_EXP_NOGATES        /* diagnostic: no transcendental work */
#define REC4M_GATES c = 0.5f * c + 0.1f * (vi + vf); h = fminf(fmaxf(0.3f * (vg + vo) + c, -1.0f), 1.0f);
#else
#define REC4M_GATES                                                                                    \
        const float ig = CSA_RCP(1.0f + CSA_EXP2(-1.44269504088896341f * vi));                         \
        const float fg = CSA_RCP(1.0f + CSA_EXP2(-1.44269504088896341f * vf));                         \
        const float og = CSA_RCP(1.0f + CSA_EXP2(-1.44269504088896341f * vo));                         \
        const float tg = fminf(CSA_EXP2(-2.88539008177792681f * vg), 1e30f);                           \
        const float gg = (1.0f - tg) * CSA_RCP(1.0f + tg);                                             \
        c = fg * c + ig * gg;                                                                          \
        const float tc = fminf(CSA_EXP2(-2.88539008177792681f * c), 1e30f);                            \
        h = og * ((1.0f - tc) * CSA_RCP(1.0f + tc));
#endif
#define MF4(ACC, K, HV, G)                                                                             \
    ACC = __builtin_amdgcn_mfma_f32_4x4x1f32(w[(K)], HV.x, ACC, 0, 0, G);                              \
    ACC##b = __builtin_amdgcn_mfma_f32_4x4x1f32(w[(K) + 1], HV.y, ACC##b, 0, 0, G);                    \
    ACC##c = __builtin_amdgcn_mfma_f32_4x4x1f32(w[(K) + 2], HV.z, ACC##c, 0, 0, G);                    \
    ACC##d = __builtin_amdgcn_mfma_f32_4x4x1f32(w[(K) + 3], HV.w, ACC##d, 0, 0, G);
#define MF4L(ACC, WV, HV, G)                                                                           \
    ACC = __builtin_amdgcn_mfma_f32_4x4x1f32(WV.x, HV.x, ACC, 0, 0, G);                                \
    ACC##b = __builtin_amdgcn_mfma_f32_4x4x1f32(WV.y, HV.y, ACC##b, 0, 0, G);                          \
    ACC##c = __builtin_amdgcn_mfma_f32_4x4x1f32(WV.z, HV.z, ACC##c, 0, 0, G);                          \
    ACC##d = __builtin_amdgcn_mfma_f32_4x4x1f32(WV.w, HV.w, ACC##d, 0, 0, G);
#define LSTM4M_STEP(T, CUR, NXT)                                                                       \
    {                                                                                                  \
        const int t_ = (T);                                                                            \
        {   /* unconditional prefetch + unconditional wait: see lstm_rec2_kernel */                     \
            const float *pn = Pb + (size_t)(t_ + 1 < L ? t_ + 1 : L - 1) * Pstep;                      \
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(NXT) : "v"(pn) : "memory");         \
        }                                                                                              \
        const float *hb = &hbuf[t_ & 1][0] + hoff;                                                     \
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accb = acc, accc = acc, accd = acc;                          \
        if (BLGP) {                                                                                    \
            _Pragma("unroll") for (int q = 0; q < NR / 16 / REC4M_EXP_MFMA_DIV; ++q) {                 \
                const f32x4 hv = *(const f32x4 *)(hb + 64 * q);                                        \
                MF4(acc, 16 * q, hv, 4)                                                                \
                MF4(acc, 16 * q + 4, hv, 5)                                                            \
                MF4(acc, 16 * q + 8, hv, 6)                                                            \
                MF4(acc, 16 * q + 12, hv, 7)                                                           \
            }                                                                                          \
            _Pragma("unroll") for (int q = 0; q < NWL / 16; ++q) {      /* the LDS-resident weight tail */ \
                const f32x4 hv = *(const f32x4 *)(hb + 64 * (NR / 16 + q));                            \
                const f32x4 w0 = wl4[(4 * q) * NT + tid], w1 = wl4[(4 * q + 1) * NT + tid];            \
                const f32x4 w2 = wl4[(4 * q + 2) * NT + tid], w3 = wl4[(4 * q + 3) * NT + tid];        \
                MF4L(acc, w0, hv, 4)                                                                   \
                MF4L(acc, w1, hv, 5)                                                                   \
                MF4L(acc, w2, hv, 6)                                                                   \
                MF4L(acc, w3, hv, 7)                                                                   \
            }                                                                                          \
        } else {                                                                                       \
            _Pragma("unroll") for (int q = 0; q < NH / 4; ++q) {                                       \
                const f32x4 hv = *(const f32x4 *)(hb + 64 * (q >> 2) + 16 * (q & 3));                   \
                MF4(acc, 4 * q, hv, 0)                                                                 \
            }                                                                                          \
        }                                                                                              \
        asm volatile("s_waitcnt vmcnt(1)" : "+v"(CUR));                                                \
        /* D rows = gate rows (i, f, g, o); the projection row is stored [i, g~, f, o] */              \
        acc = (acc + accb) + (accc + accd);            /* four interleaved k-chains: no dependent MFMA pair back to back */ \
        const float vi = acc.x + CUR.x, vf = acc.y + CUR.z, vg = acc.z + CUR.y, vo = acc.w + CUR.w;    \
        REC4M_GATES                                                                                    \
        hbuf[(t_ & 1) ^ 1][hslot] = h;                                                                 \
        if (valid) Hout[((size_t)(reverse_out ? L - 1 - t_ : t_) * B + b) * NH + u] = h;               \
        if (TRAIN && valid) {                                                                          \
            Hseq[((size_t)(t_ + 1) * B + b) * NH + u] = h;                                             \
            Cseq[((size_t)(t_ + 1) * B + b) * NH + u] = c;                                             \
            *(f32x4 *)(P + ((size_t)t_ * B + b) * (4 * NH) + u * 4) = f32x4{ig, gg, fg, og};           \
        }                                                                                              \
        LDS_BARRIER();                                                                                 \
    }
    for (int t = 0; t < L; t += 2) {
        LSTM4M_STEP(t, preA, preB)
        if (t + 1 < L) LSTM4M_STEP(t + 1, preB, preA)
    }
#undef LSTM4M_STEP
#undef MF4
#undef MF4L
#undef REC4M_GATES
}

// ------------------------------------------------------------------------------------------------
// GRU on the MATRIX pipe for LARGE batches: lstm_rec4m_kernel's mapping with three gate rows per block.  lane = 4*unit_in_wave + x
// holds W_hh[gate x of the unit][k] (x = 0, 1, 2: r, z, n; x = 3: zeros -- a quarter of every 4x4x1 MFMA is padding) as A operand
// and h_{t-1}[k] of column x as B operand; its D registers are W_hr h, W_hz h, W_hn h of (unit, column x), so
//   r = sigma(x_r + D.x), z = sigma(x_z + D.y), n = tanh(x_n + r (D.z + b_hn)), h = (1 - z) n + z h
// happen in the lane, no cross-lane traffic.  P rows are [r, z, n] per unit, unpadded: one 12-byte load per step.
// Same arithmetic as gru_rec2_kernel up to the order of the k-sum.
// TRAIN (round 3): the training forward of the GRU models at shard size -- P rows are the 4-padded rows of the training buffers, the
// lane ends a step with [r, z, n, W_hn h + b_hn] of its (unit, column) and writes them IN PLACE over the row it consumed, h_t also to
// Hseq (L+1 slots, slot 0 = the initial state): what gru_rec2_kernel<NH, true> saves and gru_bwd_rec_kernel reads.
template <int NH, bool BLGP, bool TRAIN = false>
__global__ __launch_bounds__(NH * 4, (NH / 16 + 3) / 4) void gru_rec4m_kernel(
    const float *__restrict__ Wk, const float *__restrict__ bhn, const float *P,
    const float *__restrict__ h0, float *__restrict__ Hout, int B, int L, int reverse_out, float *Pw = nullptr, float *__restrict__ Hseq = nullptr)
{
    constexpr int PS = TRAIN ? 4 : 3;
    constexpr int NT = NH * 4;
    static_assert(NH % 16 == 0, "nh must be a multiple of 16");
    __shared__ __attribute__((aligned(16))) float hbuf[2][4 * NH];      // layout: see lstm_rec4m_kernel

    const int tid = threadIdx.x, u = tid >> 2, x = tid & 3, lane = tid & 63;
    int b = 4 * blockIdx.x + x;
    const bool valid = b < B;
    if (!valid) b = B - 1;

    float w[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) w[k] = Wk[(size_t)k * NT + tid];

    const float bn = bhn[u];
    float h = h0[(size_t)b * NH + u];
    const int hslot = (u >> 4) * 64 + ((((u & 15) >> 2) * 4 + x) << 2) + (u & 3);
    hbuf[0][hslot] = h;
    if (TRAIN && valid) Hseq[(size_t)b * NH + u] = h;
    const float *Pb = P + (size_t)b * (PS * NH) + u * PS;
    const size_t Pstep = (size_t)B * (PS * NH);
    typedef typename std::conditional<TRAIN, f32x4, f32x3>::type prow;
    prow preA, preB;
    preA.x = Pb[0]; preA.y = Pb[1]; preA.z = Pb[2];
    preB = preA;
    const int hoff = BLGP ? (((lane >> 4) * 4 + x) << 2) : (x << 2);
    __syncthreads();

#define MF4(ACC, K, HV, G)                                                                             \
    ACC = __builtin_amdgcn_mfma_f32_4x4x1f32(w[(K)], HV.x, ACC, 0, 0, G);                              \
    ACC##b = __builtin_amdgcn_mfma_f32_4x4x1f32(w[(K) + 1], HV.y, ACC##b, 0, 0, G);                    \
    ACC##c = __builtin_amdgcn_mfma_f32_4x4x1f32(w[(K) + 2], HV.z, ACC##c, 0, 0, G);                    \
    ACC##d = __builtin_amdgcn_mfma_f32_4x4x1f32(w[(K) + 3], HV.w, ACC##d, 0, 0, G);
#define GRU4M_STEP(T, CUR, NXT)                                                                        \
    {                                                                                                  \
        const int t_ = (T);                                                                            \
        {   /* unconditional prefetch + unconditional wait: see lstm_rec2_kernel */                     \
            const float *pn = Pb + (size_t)(t_ + 1 < L ? t_ + 1 : L - 1) * Pstep;                      \
            if (TRAIN) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(NXT) : "v"(pn) : "memory"); \
            else asm volatile("global_load_dwordx3 %0, %1, off" : "=&v"(NXT) : "v"(pn) : "memory");    \
        }                                                                                              \
        const float *hb = &hbuf[t_ & 1][0] + hoff;                                                     \
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accb = acc, accc = acc, accd = acc;                          \
        if (BLGP) {                                                                                    \
            _Pragma("unroll") for (int q = 0; q < NH / 16; ++q) {                                      \
                const f32x4 hv = *(const f32x4 *)(hb + 64 * q);                                        \
                MF4(acc, 16 * q, hv, 4)                                                                \
                MF4(acc, 16 * q + 4, hv, 5)                                                            \
                MF4(acc, 16 * q + 8, hv, 6)                                                            \
                MF4(acc, 16 * q + 12, hv, 7)                                                           \
            }                                                                                          \
        } else {                                                                                       \
            _Pragma("unroll") for (int q = 0; q < NH / 4; ++q) {                                       \
                const f32x4 hv = *(const f32x4 *)(hb + 64 * (q >> 2) + 16 * (q & 3));                   \
                MF4(acc, 4 * q, hv, 0)                                                                 \
            }                                                                                          \
        }                                                                                              \
        asm volatile("s_waitcnt vmcnt(1)" : "+v"(CUR));                                                \
        acc = (acc + accb) + (accc + accd);                                                            \
        const float rg = CSA_RCP(1.0f + CSA_EXP2(-1.44269504088896341f * (acc.x + CUR.x)));            \
        const float zg = CSA_RCP(1.0f + CSA_EXP2(-1.44269504088896341f * (acc.y + CUR.y)));            \
        const float tn = fminf(CSA_EXP2(-2.88539008177792681f * (CUR.z + rg * (acc.z + bn))), 1e30f);  \
        const float n = (1.0f - tn) * CSA_RCP(1.0f + tn);                                              \
        h = (1.0f - zg) * n + zg * h;                                                                  \
        hbuf[(t_ & 1) ^ 1][hslot] = h;                                                                 \
        if (valid) Hout[((size_t)(reverse_out ? L - 1 - t_ : t_) * B + b) * NH + u] = h;               \
        if (TRAIN && valid) {                                                                          \
            Hseq[((size_t)(t_ + 1) * B + b) * NH + u] = h;                                             \
            *(f32x4 *)(Pw + ((size_t)t_ * B + b) * (4 * NH) + u * 4) = f32x4{rg, zg, n, acc.z + bn};   \
        }                                                                                              \
        LDS_BARRIER();                                                                                 \
    }
    for (int t = 0; t < L; t += 2) {
        GRU4M_STEP(t, preA, preB)
        if (t + 1 < L) GRU4M_STEP(t + 1, preB, preA)
    }
#undef GRU4M_STEP
#undef MF4
}

// ------------------------------------------------------------------------------------------------
// GRU kernel, second generation (inference): the lstm_rec2_kernel design with three accumulator slots.
// Slot order per lane group (host packing, gru2_pack_weights): p<2 holds [r, hn, z], p>=2 holds [z, hn, r], so after
//   r[s] = acc[s].x + xor1(acc[s].y);  v0 = r[0] + xor2(r[2]);  v1 = r[1] + xor2(r[1])
// every lane has W_hn h (v1) and the (r,z)-gate sum of ITS group (v0): p<2 the reset gate, p>=2 the update gate.
// 8 DPP adds per step (the all-reduce of rec_kernel<NH,3> needs 12 plus a select), 96 packed FMAs as before.
// P rows are [r, z, n] per unit: lane group g reads the 8-byte pair at offset g -> (r,z) or (z,n); the p>=2 lanes own
// h_t: they receive r over one DPP move, form n = tanh(x_n + r (W_hn h + b_hn)) and h = (1-z) n + z h.
// TRAIN (round 3: the training forward of the GRU models, which ran on the first-generation rec_kernel<NH,3,true>): P rows are
// the 4-padded rows of the training buffers; the owner lane ends a step with all of [r, z, n, W_hn h + b_hn] of its (unit,
// column) and writes them IN PLACE over the row it consumed, h_t also goes to Hseq (L+1 slots, slot 0 = h_init) -- what
// gru_bwd_rec_kernel reads.
template <int NH, bool TRAIN = false>
__global__ __launch_bounds__(NH * 4, (NH / 16 + 3) / 4) void gru_rec2_kernel(
    const f32x4 *__restrict__ Wp4, const float *__restrict__ bhn, const float *P,
    const float *__restrict__ h0, float *__restrict__ Hout, int B, int L, int reverse_out,
    float *Pw = nullptr, float *__restrict__ Hseq = nullptr)
{
    constexpr int PS = TRAIN ? 4 : 3;     // floats per unit of a P row
    constexpr int NT = NH * 4;
    constexpr int KC = NH / 4;
    constexpr int CH = 2 * KC + 4;
    constexpr int CPY = 4 * CH;
    constexpr int KR = KC / 2;
    static_assert(KC % 4 == 0, "nh must be a multiple of 16");
    __shared__ __attribute__((aligned(16))) float hbuf[2][2 * CPY];

    const int tid = threadIdx.x, u = tid >> 2, p = tid & 3, col = p & 1, grp = p >> 1;
    int b = 2 * blockIdx.x + col;
    const bool valid = b < B;
    if (!valid) b = B - 1;
    const bool owner = grp == 1;

    f32x2 w[3][KR];
#pragma unroll
    for (int i = 0; i < 3 * KC / 4; ++i) {
        const f32x4 v = Wp4[(size_t)i * NT + tid];
        const int s = (4 * i) / KC, kk = (4 * i) % KC;
        w[s][kk / 2] = f32x2{v.x, v.y};
        w[s][kk / 2 + 1] = f32x2{v.z, v.w};
    }
    const float bn = bhn[u];
    float h = h0[(size_t)b * NH + u];
    const int slotN = 2 * u + col + 4 * (u / KC);
    const int slotS = CPY + 2 * u + (1 - col) + 4 * (u / KC);
    if (owner) { hbuf[0][slotN] = h; hbuf[0][slotS] = h; }
    if (TRAIN && owner && valid) Hseq[(size_t)b * NH + u] = h;

    const float *Pb = P + (size_t)b * (PS * NH) + u * PS + grp;    // rows are [r, z, n(, pad)] per unit: 4-byte-aligned pairs
    const size_t Pstep = (size_t)B * (PS * NH);
    const int rdoff = col * CPY + p * CH;
    f32x2 preA = f32x2{Pb[0], Pb[1]}, preB = preA;
    __syncthreads();

#define GRU2_STEP(T, CUR, NXT)                                                                     \
    {                                                                                              \
        const int t_ = (T);                                                                        \
        {   /* unconditional prefetch + unconditional wait: see lstm_rec2_kernel */                 \
            const float *pn = Pb + (size_t)(t_ + 1 < L ? t_ + 1 : L - 1) * Pstep;                  \
            asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(NXT) : "v"(pn) : "memory");     \
        }                                                                                          \
        const f32x4 *hp = (const f32x4 *)&hbuf[t_ & 1][rdoff];                                     \
        f32x2 acc[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};                                       \
        _Pragma("unroll") for (int j = 0; j < KR; ++j) {                                           \
            const f32x4 hv = hp[j];                                                                \
            const f32x2 ha = {hv.x, hv.y}, hb = {hv.z, hv.w};                                      \
            _Pragma("unroll") for (int s = 0; s < 3; ++s) PK_FMA_LO(acc[s], w[s][j], ha);          \
            _Pragma("unroll") for (int s = 0; s < 3; ++s) PK_FMA_HI(acc[s], w[s][j], hb);          \
        }                                                                                          \
        float r[3];                                                                                \
        _Pragma("unroll") for (int s = 0; s < 3; ++s) r[s] = acc[s].x + dpp_xor1(acc[s].y);        \
        asm volatile("s_waitcnt vmcnt(1)" : "+v"(CUR));                                            \
        const float v0 = r[0] + dpp_xor2(r[2]) + CUR.x;                                            \
        const float hn = r[1] + dpp_xor2(r[1]) + bn;                                               \
        const float g0 = CSA_RCP(1.0f + CSA_EXP2(-1.44269504088896341f * v0)); \
        const float rr = dpp_xor2(g0);                /* the reset gate arrives at the lane that owns h */ \
        const float tn = fminf(CSA_EXP2(-2.88539008177792681f * (CUR.y + rr * hn)), 1e30f); \
        const float n = (1.0f - tn) * CSA_RCP(1.0f + tn);                            \
        h = (1.0f - g0) * n + g0 * h;                                                              \
        if (owner) {                                                                               \
            hbuf[(t_ & 1) ^ 1][slotN] = h;                                                         \
            hbuf[(t_ & 1) ^ 1][slotS] = h;                                                         \
            if (valid) Hout[((size_t)(reverse_out ? L - 1 - t_ : t_) * B + b) * NH + u] = h;       \
            if (TRAIN && valid) {                                                                  \
                Hseq[((size_t)(t_ + 1) * B + b) * NH + u] = h;                                     \
                *(f32x4 *)(Pw + ((size_t)t_ * B + b) * (4 * NH) + u * 4) = f32x4{rr, g0, n, hn}; \
            }                                                                                      \
        }                                                                                          \
        LDS_BARRIER();                                                                             \
    }
    for (int t = 0; t < L; t += 2) {
        GRU2_STEP(t, preA, preB)
        if (t + 1 < L) GRU2_STEP(t + 1, preB, preA)
    }
#undef GRU2_STEP
}

// ------------------------------------------------------------------------------------------------
// LSTM kernel, ONE column per workgroup: the latency variant for small batches (B <= 256: every column gets its own
// CU).  Same register-stationary weights (4 gate rows x nh/4 k-values per lane), but the packed FMA pairs two
// consecutive k of the SAME column (acc.x + acc.y at the end), so a step costs 64 v_pk_fma_f32 per lane instead of
// 128, half the LDS reads and four instead of six transcendental pairs.  The sum over the four k-quarters is a
// reduce-scatter of 3 DPP adds that leaves lane p of a quad with the complete pre-activation of gate p of its unit
// (slot order per lane sigma_p = [p, p^1, p^2, p^3] over [i, g~, f, o], baked into the host packing); lane p applies
// that gate's activation, three DPP moves bring i*g~, f and o to every lane, and c_t / h_t are kept redundantly by all
// four lanes of the quad.  P rows are [i, g~, f, o] per unit: lane p reads one float.
template <int NH>
__global__ __launch_bounds__(NH * 4, (NH / 16 + 3) / 4) void lstm_rec1_kernel(
    const f32x4 *__restrict__ Wp4, const float *__restrict__ P,
    const float *__restrict__ h0, const float *__restrict__ c0, float *__restrict__ Hout,
    int B, int L, int reverse_out)
{
    constexpr int NT = NH * 4;
    constexpr int KC = NH / 4;
    constexpr int CH = KC + 4;          // floats per k-quarter incl. one 16-B pad slot (4 distinct addresses -> 4 bank groups)
    static_assert(KC % 4 == 0, "nh must be a multiple of 16");
    __shared__ __attribute__((aligned(16))) float hbuf[2][4 * CH];

    const int tid = threadIdx.x, u = tid >> 2, p = tid & 3;
    const int b = blockIdx.x;

    f32x2 w[4][KC / 2];
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        const f32x4 v = Wp4[(size_t)i * NT + tid];
        const int s = (4 * i) / KC, kk = (4 * i) % KC;
        w[s][kk / 2] = f32x2{v.x, v.y};
        w[s][kk / 2 + 1] = f32x2{v.z, v.w};
    }
    // gate p: sigmoid for i (0), f (2), o (3); tanh for g~ (1) -- both as (1 - nb*t)/(1 + t), t = exp2(k*x)
    const float kact = p == 1 ? -2.88539008177792681f : -1.44269504088896341f;
    const float nb = p == 1 ? 1.0f : 0.0f;

    float h = h0[(size_t)b * NH + u];
    float c = c0[(size_t)b * NH + u];
    const int hslot = u + 4 * (u / KC);
    if (p == 0) hbuf[0][hslot] = h;
    asm volatile("" : "+v"(c), "+v"(h));
    const float *Pb = P + (size_t)b * (4 * NH) + u * 4 + p;
    const size_t Pstep = (size_t)B * (4 * NH);
    float preA = Pb[0], preB = preA;
    __syncthreads();

#define LSTM1_STEP(T, CUR, NXT)                                                                    \
    {                                                                                              \
        const int t_ = (T);                                                                        \
        {   /* unconditional prefetch + unconditional wait: see lstm_rec2_kernel */                 \
            const float *pn = Pb + (size_t)(t_ + 1 < L ? t_ + 1 : L - 1) * Pstep;                  \
            asm volatile("global_load_dword %0, %1, off" : "=&v"(NXT) : "v"(pn) : "memory");       \
        }                                                                                          \
        const f32x4 *hp = (const f32x4 *)&hbuf[t_ & 1][p * CH];                                    \
        f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};                           \
        _Pragma("unroll") for (int j = 0; j < KC / 4; ++j) {                                       \
            const f32x4 hv = hp[j];                                                                \
            const f32x2 ha = {hv.x, hv.y}, hb = {hv.z, hv.w};                                      \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[s] = __builtin_elementwise_fma(w[s][2 * j], ha, acc[s]);     \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[s] = __builtin_elementwise_fma(w[s][2 * j + 1], hb, acc[s]); \
        }                                                                                          \
        float sm[4];                                                                               \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) sm[s] = acc[s].x + acc[s].y;                 \
        const float r0 = sm[0] + dpp_xor1(sm[1]);                                                  \
        const float r1 = sm[2] + dpp_xor1(sm[3]);                                                  \
        asm volatile("s_waitcnt vmcnt(1)" : "+v"(CUR));                                            \
        const float v = r0 + dpp_xor2(r1) + CUR;                                                   \
        const float te = fminf(CSA_EXP2(kact * v), 1e30f);                           \
        const float a = (1.0f - nb * te) * CSA_RCP(1.0f + te);     /* gate p of unit u */ \
        const float ig = a * dpp_xor1(a);                                        /* lanes 0,1: i*g~ */ \
        const float igq = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ig), 0x00, 0xF, 0xF, true)); \
        const float fq = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xAA, 0xF, 0xF, true));  \
        const float oq = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xFF, 0xF, 0xF, true));  \
        c = fq * c + igq;                                                                          \
        const float tc = fminf(CSA_EXP2(-2.88539008177792681f * c), 1e30f);          \
        h = oq * ((1.0f - tc) * CSA_RCP(1.0f + tc));                                 \
        if (p == 0) {                                                                              \
            hbuf[(t_ & 1) ^ 1][hslot] = h;                                                         \
            Hout[((size_t)(reverse_out ? L - 1 - t_ : t_) * B + b) * NH + u] = h;                  \
        }                                                                                          \
        LDS_BARRIER();                                                                             \
    }
    for (int t = 0; t < L; t += 2) {
        LSTM1_STEP(t, preA, preB)
        if (t + 1 < L) LSTM1_STEP(t + 1, preB, preA)
    }
#undef LSTM1_STEP
}

// GRU, one column per workgroup (small batches): lane (u, p) holds the three gate rows of unit u for its k-quarter as
// k-pairs (48 v_pk_fma_f32 per step), the quad all-reduces the three sums (6 DPP adds) and every lane of the quad
// evaluates the cell redundantly.  P rows are [r, z, n] per unit (pack.h), b_hn is kept apart as in rec_kernel.
template <int NH>
__global__ __launch_bounds__(NH * 4, (NH / 16 + 3) / 4) void gru_rec1_kernel(
    const f32x4 *__restrict__ Wp4, const float *__restrict__ bhn, const float *__restrict__ P,
    const float *__restrict__ h0, float *__restrict__ Hout, int B, int L, int reverse_out)
{
    constexpr int NT = NH * 4;
    constexpr int KC = NH / 4;
    constexpr int CH = KC + 4;
    static_assert(KC % 4 == 0, "nh must be a multiple of 16");
    __shared__ __attribute__((aligned(16))) float hbuf[2][4 * CH];
    const int tid = threadIdx.x, u = tid >> 2, p = tid & 3;
    const int b = blockIdx.x;
    f32x2 w[3][KC / 2];
#pragma unroll
    for (int i = 0; i < 3 * KC / 4; ++i) {
        const f32x4 v = Wp4[(size_t)i * NT + tid];
        const int g = (4 * i) / KC, kk = (4 * i) % KC;
        w[g][kk / 2] = f32x2{v.x, v.y};
        w[g][kk / 2 + 1] = f32x2{v.z, v.w};
    }
    const float bn = bhn[u];
    float h = h0[(size_t)b * NH + u];
    const int hslot = u + 4 * (u / KC);
    if (p == 0) hbuf[0][hslot] = h;
    asm volatile("" : "+v"(h));
    const float *Pb = P + (size_t)b * (3 * NH) + u * 3;           // rows are [r, z, n] per unit, unpadded
    const size_t Pstep = (size_t)B * (3 * NH);
    f32x3 preA = {Pb[0], Pb[1], Pb[2]}, preB = preA;
    __syncthreads();
#define GRU1_STEP(T, CUR, NXT)                                                                     \
    {                                                                                              \
        const int t_ = (T);                                                                        \
        {   /* unconditional prefetch + unconditional wait: see lstm_rec2_kernel */                 \
            const float *pn = Pb + (size_t)(t_ + 1 < L ? t_ + 1 : L - 1) * Pstep;                  \
            asm volatile("global_load_dwordx3 %0, %1, off" : "=&v"(NXT) : "v"(pn) : "memory");     \
        }                                                                                          \
        const f32x4 *hp = (const f32x4 *)&hbuf[t_ & 1][p * CH];                                    \
        f32x2 acc[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};                                       \
        _Pragma("unroll") for (int j = 0; j < KC / 4; ++j) {                                       \
            const f32x4 hv = hp[j];                                                                \
            const f32x2 ha = {hv.x, hv.y}, hb = {hv.z, hv.w};                                      \
            _Pragma("unroll") for (int g = 0; g < 3; ++g) acc[g] = __builtin_elementwise_fma(w[g][2 * j], ha, acc[g]);     \
            _Pragma("unroll") for (int g = 0; g < 3; ++g) acc[g] = __builtin_elementwise_fma(w[g][2 * j + 1], hb, acc[g]); \
        }                                                                                          \
        const float sr = quad_sum(acc[0].x + acc[0].y), sz = quad_sum(acc[1].x + acc[1].y);        \
        const float sn = quad_sum(acc[2].x + acc[2].y);                                            \
        asm volatile("s_waitcnt vmcnt(1)" : "+v"(CUR));                                            \
        const float r = sigmoid_f(CUR.x + sr);                                                     \
        const float z = sigmoid_f(CUR.y + sz);                                                     \
        const float n = tanh_f(CUR.z + r * (sn + bn));                                             \
        h = (1.0f - z) * n + z * h;                                                                \
        if (p == 0) {                                                                              \
            hbuf[(t_ & 1) ^ 1][hslot] = h;                                                         \
            Hout[((size_t)(reverse_out ? L - 1 - t_ : t_) * B + b) * NH + u] = h;                  \
        }                                                                                          \
        LDS_BARRIER();                                                                             \
    }
    for (int t = 0; t < L; t += 2) {
        GRU1_STEP(t, preA, preB)
        if (t + 1 < L) GRU1_STEP(t + 1, preB, preA)
    }
#undef GRU1_STEP
}

int launch_rec1_gru(int nh, const float *whh_packed, const float *bhn, const float *P, const float *h0, float *Hout, int B, int L,
                    int reverse_out, hipStream_t s)
{
    // the GRU packing of rec_pack_weights (thread-major: [g][kk] for unit u, k-quarter p) is exactly what this kernel reads
    const dim3 grid(B), block(nh * 4);
    switch (nh) {
    case 64:  hipLaunchKernelGGL((gru_rec1_kernel<64>), grid, block, 0, s, (const f32x4 *)whh_packed, bhn, P, h0, Hout, B, L, reverse_out); break;
    case 96:  hipLaunchKernelGGL((gru_rec1_kernel<96>), grid, block, 0, s, (const f32x4 *)whh_packed, bhn, P, h0, Hout, B, L, reverse_out); break;
    case 112: hipLaunchKernelGGL((gru_rec1_kernel<112>), grid, block, 0, s, (const f32x4 *)whh_packed, bhn, P, h0, Hout, B, L, reverse_out); break;
    case 128: hipLaunchKernelGGL((gru_rec1_kernel<128>), grid, block, 0, s, (const f32x4 *)whh_packed, bhn, P, h0, Hout, B, L, reverse_out); break;
    case 144: hipLaunchKernelGGL((gru_rec1_kernel<144>), grid, block, 0, s, (const f32x4 *)whh_packed, bhn, P, h0, Hout, B, L, reverse_out); break;
    default:
        csa_set_error_msg("rec1(GRU): hidden size not supported (64, 96, 128, 144)");
        return CSA_ERR_UNSUPPORTED;
    }
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// lane p, slot s holds gate (p ^ s) of [i, g~, f, o]  ->  PyTorch gate rows (i, f, g, o)
static void lstm1_pack_weights(int nh, const float *w_hh, float *packed)
{
    static const int pos2gate[4] = {0, 2, 1, 3};     // position in [i, g~, f, o] -> PyTorch gate index
    const int NT = nh * 4, KC = nh / 4;
    for (int tid = 0; tid < NT; ++tid) {
        const int u = tid >> 2, p = tid & 3;
        for (int idx = 0; idx < 4 * KC; ++idx) {
            const int s = idx / KC, kk = idx % KC, g = pos2gate[p ^ s];
            const int i = idx / 4, e = idx % 4;
            packed[((size_t)i * NT + tid) * 4 + e] = w_hh[(size_t)(g * nh + u) * nh + p * KC + kk];
        }
    }
}
void rec1_pack_weights(int nh, const float *w_hh, float *packed) { lstm1_pack_weights(nh, w_hh, packed); }

int launch_rec1(int nh, const float *whh_packed1, const float *P, const float *h0, const float *c0, float *Hout, int B, int L,
                int reverse_out, hipStream_t s)
{
    const dim3 grid(B), block(nh * 4);
    switch (nh) {
    case 64:  hipLaunchKernelGGL((lstm_rec1_kernel<64>), grid, block, 0, s, (const f32x4 *)whh_packed1, P, h0, c0, Hout, B, L, reverse_out); break;
    case 96:  hipLaunchKernelGGL((lstm_rec1_kernel<96>), grid, block, 0, s, (const f32x4 *)whh_packed1, P, h0, c0, Hout, B, L, reverse_out); break;
    case 128: hipLaunchKernelGGL((lstm_rec1_kernel<128>), grid, block, 0, s, (const f32x4 *)whh_packed1, P, h0, c0, Hout, B, L, reverse_out); break;
    case 144: hipLaunchKernelGGL((lstm_rec1_kernel<144>), grid, block, 0, s, (const f32x4 *)whh_packed1, P, h0, c0, Hout, B, L, reverse_out); break;
    default:
        csa_set_error_msg("rec1: hidden size not supported (64, 96, 128, 144)");
        return CSA_ERR_UNSUPPORTED;
    }
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// slot -> PyTorch gate index (i,f,g,o = 0,1,2,3) for lane groups p<2 and p>=2
static const int kLstm2Slot[2][4] = {{0, 2, 1, 3}, {1, 3, 0, 2}};

static void lstm2_pack_weights(int nh, const float *w_hh, float *packed)
{
    const int NT = nh * 4, KC = nh / 4;
    for (int tid = 0; tid < NT; ++tid) {
        const int u = tid >> 2, p = tid & 3, grp = p >> 1;
        for (int idx = 0; idx < 4 * KC; ++idx) {
            const int s = idx / KC, kk = idx % KC, g = kLstm2Slot[grp][s];
            const int i = idx / 4, e = idx % 4;
            packed[((size_t)i * NT + tid) * 4 + e] = w_hh[(size_t)(g * nh + u) * nh + p * KC + kk];
        }
    }
}

// second-generation GRU kernel: slot s of lane group grp holds PyTorch gate kGru2Slot[grp][s] (r = 0, z = 1, n = 2)
static const int kGru2Slot[2][3] = {{0, 2, 1}, {1, 2, 0}};
void gru2_pack_weights(int nh, const float *w_hh, float *packed)
{
    const int NT = nh * 4, KC = nh / 4;
    for (int tid = 0; tid < NT; ++tid) {
        const int u = tid >> 2, p = tid & 3, grp = p >> 1;
        for (int idx = 0; idx < 3 * KC; ++idx) {
            const int s = idx / KC, kk = idx % KC, g = kGru2Slot[grp][s];
            const int i = idx / 4, e = idx % 4;
            packed[((size_t)i * NT + tid) * 4 + e] = w_hh[(size_t)(g * nh + u) * nh + p * KC + kk];
        }
    }
}

int launch_rec2_gru(int nh, const float *whh_g2, const float *bhn, const float *P, const float *h0, float *Hout, int B, int L,
                    int reverse_out, hipStream_t s)
{
    const dim3 grid((B + 1) / 2), block(nh * 4);
    switch (nh) {
    case 64:  hipLaunchKernelGGL((gru_rec2_kernel<64>), grid, block, 0, s, (const f32x4 *)whh_g2, bhn, P, h0, Hout, B, L, reverse_out); break;
    case 96:  hipLaunchKernelGGL((gru_rec2_kernel<96>), grid, block, 0, s, (const f32x4 *)whh_g2, bhn, P, h0, Hout, B, L, reverse_out); break;
    case 112: hipLaunchKernelGGL((gru_rec2_kernel<112>), grid, block, 0, s, (const f32x4 *)whh_g2, bhn, P, h0, Hout, B, L, reverse_out); break;
    case 128: hipLaunchKernelGGL((gru_rec2_kernel<128>), grid, block, 0, s, (const f32x4 *)whh_g2, bhn, P, h0, Hout, B, L, reverse_out); break;
    case 144: hipLaunchKernelGGL((gru_rec2_kernel<144>), grid, block, 0, s, (const f32x4 *)whh_g2, bhn, P, h0, Hout, B, L, reverse_out); break;
    default:
        csa_set_error_msg("rec2(GRU): hidden size not supported (64, 96, 128, 144)");
        return CSA_ERR_UNSUPPORTED;
    }
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

size_t rec_packed_floats(int use_lstm, int nh) { return (size_t)(use_lstm ? 4 : 3) * nh * nh; }

void rec_pack_weights(int use_lstm, int nh, const float *w_hh, float *packed)
{
    if (use_lstm) { lstm2_pack_weights(nh, w_hh, packed); return; }
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

static int rec4_min_batch();
// matrix-pipe kernel: Wk[k*NT + tid] = W_hh[gate (tid & 3) of unit (tid >> 2)][k], PyTorch gate rows (i, f, g, o)
void rec4m_pack_weights(int nh, const float *w_hh, float *packed)
{
    const int NT = nh * 4;
    for (int k = 0; k < nh; ++k)
        for (int tid = 0; tid < NT; ++tid)
            packed[(size_t)k * NT + tid] = w_hh[(size_t)((tid & 3) * nh + (tid >> 2)) * nh + k];
}

// which large-batch LSTM kernel: CSA_REC4_KERNEL = mfma (default: 4x4x1 matrix pipe, BLGP operand broadcast), mfma_noblgp,
// valu (the packed-FMA four-column kernel)
static int rec4_variant()
{
    static const int v = [] {
        const char *e = getenv("CSA_REC4_KERNEL");
        if (!e) return 2;
        return !strcmp(e, "valu") ? 0 : (!strcmp(e, "mfma_noblgp") ? 1 : 2);
    }();
    return v;
}

int launch_rec4m(int nh, const float *whh_m, const float *P, const float *h0, const float *c0, float *Hout, int B, int L,
                 int reverse_out, hipStream_t s)
{
    const dim3 grid((B + 3) / 4), block(nh * 4);
    const bool blgp = rec4_variant() == 2;
#define L4M(NHv)                                                                                                             \
    if (blgp) hipLaunchKernelGGL((lstm_rec4m_kernel<NHv, true>), grid, block, 0, s, whh_m, (float *)P, h0, c0, Hout, B, L, reverse_out, (float *)nullptr, (float *)nullptr);   \
    else hipLaunchKernelGGL((lstm_rec4m_kernel<NHv, false>), grid, block, 0, s, whh_m, (float *)P, h0, c0, Hout, B, L, reverse_out, (float *)nullptr, (float *)nullptr);
    switch (nh) {
    case 64: L4M(64) break;
    case 96: L4M(96) break;
    case 128: L4M(128) break;
    case 144: {
        constexpr size_t shm = (size_t)(REC4M_NWL144 / 4) * 144 * 4 * sizeof(f32x4);
        auto kern = lstm_rec4m_kernel<144, true, false, REC4M_NWL144>;
        CSA_SET_DYN_LDS_ONCE(kern, shm);
        hipLaunchKernelGGL(kern, grid, block, shm, s, whh_m, (float *)P, h0, c0, Hout, B, L, reverse_out, (float *)nullptr, (float *)nullptr);
        break;
    }
    default:
        csa_set_error_msg("rec4m: hidden size not supported (64, 96, 128, 144)");
        return CSA_ERR_UNSUPPORTED;
    }
#undef L4M
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// training forward on the matrix pipe (from CSA_REC4_MIN_BATCH columns): gates saved in place over P, c / h sequences (L+1 slots)
int launch_rec4m_train(int nh, const float *whh_m, float *P, const float *h0, const float *c0, float *Hout, int B, int L,
                       int reverse_out, float *Hseq, float *Cseq, hipStream_t s)
{
    const dim3 grid((B + 3) / 4), block(nh * 4);
    switch (nh) {
    case 64:  hipLaunchKernelGGL((lstm_rec4m_kernel<64, true, true>), grid, block, 0, s, whh_m, P, h0, c0, Hout, B, L, reverse_out, Hseq, Cseq); break;
    case 96:  hipLaunchKernelGGL((lstm_rec4m_kernel<96, true, true>), grid, block, 0, s, whh_m, P, h0, c0, Hout, B, L, reverse_out, Hseq, Cseq); break;
    case 128: hipLaunchKernelGGL((lstm_rec4m_kernel<128, true, true>), grid, block, 0, s, whh_m, P, h0, c0, Hout, B, L, reverse_out, Hseq, Cseq); break;
    case 144: {
        constexpr size_t shm = (size_t)(REC4M_NWL144 / 4) * 144 * 4 * sizeof(f32x4);
        auto kern = lstm_rec4m_kernel<144, true, true, REC4M_NWL144>;
        CSA_SET_DYN_LDS_ONCE(kern, shm);
        hipLaunchKernelGGL(kern, grid, block, shm, s, whh_m, P, h0, c0, Hout, B, L, reverse_out, Hseq, Cseq);
        break;
    }
    default:
        csa_set_error_msg("rec4m(train): hidden size not supported (64, 96, 128, 144)");
        return CSA_ERR_UNSUPPORTED;
    }
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
bool rec4m_selected(int use_lstm, int nh, int B) { return use_lstm && (nh <= 128 || (nh == 144 && rec4_variant() == 2)) && rec4_variant() != 0 && B >= rec4_min_batch(); }

// GRU matrix-pipe kernel: Wk[k*NT + tid] = W_hh[gate (tid & 3) of unit (tid >> 2)][k] for the PyTorch gate rows (r, z, n); the
// fourth row of every block is zero
void gru4m_pack_weights(int nh, const float *w_hh, float *packed)
{
    const int NT = nh * 4;
    for (int k = 0; k < nh; ++k)
        for (int tid = 0; tid < NT; ++tid)
            packed[(size_t)k * NT + tid] = (tid & 3) < 3 ? w_hh[(size_t)((tid & 3) * nh + (tid >> 2)) * nh + k] : 0.0f;
}
int launch_rec4m_gru(int nh, const float *whh_m, const float *bhn, const float *P, const float *h0, float *Hout, int B, int L,
                     int reverse_out, hipStream_t s)
{
    const dim3 grid((B + 3) / 4), block(nh * 4);
    const bool blgp = rec4_variant() != 1;
#define G4M(NHv)                                                                                                                 \
    if (blgp) hipLaunchKernelGGL((gru_rec4m_kernel<NHv, true>), grid, block, 0, s, whh_m, bhn, P, h0, Hout, B, L, reverse_out);  \
    else hipLaunchKernelGGL((gru_rec4m_kernel<NHv, false>), grid, block, 0, s, whh_m, bhn, P, h0, Hout, B, L, reverse_out);
    switch (nh) {
    case 64: G4M(64) break;
    case 96: G4M(96) break;
    case 112: G4M(112) break;
    case 128: G4M(128) break;
    default:
        csa_set_error_msg("gru_rec4m: hidden size not supported (64, 96, 112, 128)");
        return CSA_ERR_UNSUPPORTED;
    }
#undef G4M
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
// training forward on the matrix pipe (gates saved in place over the 4-padded rows, h sequence into Hseq)
int launch_rec4m_train_gru(int nh, const float *whh_m, const float *bhn, float *P, const float *h0, float *Hout, int B, int L,
                           int reverse_out, float *Hseq, hipStream_t s)
{
    const dim3 grid((B + 3) / 4), block(nh * 4);
    switch (nh) {
    case 64:  hipLaunchKernelGGL((gru_rec4m_kernel<64, true, true>), grid, block, 0, s, whh_m, bhn, P, h0, Hout, B, L, reverse_out, P, Hseq); break;
    case 96:  hipLaunchKernelGGL((gru_rec4m_kernel<96, true, true>), grid, block, 0, s, whh_m, bhn, P, h0, Hout, B, L, reverse_out, P, Hseq); break;
    case 112: hipLaunchKernelGGL((gru_rec4m_kernel<112, true, true>), grid, block, 0, s, whh_m, bhn, P, h0, Hout, B, L, reverse_out, P, Hseq); break;
    case 128: hipLaunchKernelGGL((gru_rec4m_kernel<128, true, true>), grid, block, 0, s, whh_m, bhn, P, h0, Hout, B, L, reverse_out, P, Hseq); break;
    default:
        csa_set_error_msg("gru_rec4m(train): hidden size not supported (64, 96, 112, 128)");
        return CSA_ERR_UNSUPPORTED;
    }
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
// the GRU flavour of rec4m_selected (CSA_GRU4M=0 keeps the two-column kernel)
bool gru4m_selected(int nh, int B)
{
    static const int on = getenv("CSA_GRU4M") ? atoi(getenv("CSA_GRU4M")) : 1;
    return on && nh <= 128 && nh % 16 == 0 && rec4_variant() != 0 && B >= rec4_min_batch();
}

// smallest batch of a launch that uses the four-column kernel (env CSA_REC4_MIN_BATCH; 0 disables)
static int rec4_min_batch()
{
    // measured (tools/rec_bench): one round of four-column workgroups costs 100 us (matrix pipe) / 130 us (packed FMA), one round of
    // two-column workgroups 57 us: from 513 columns the two-column kernel needs a second round, so four columns win
    static const int v = getenv("CSA_REC4_MIN_BATCH") ? atoi(getenv("CSA_REC4_MIN_BATCH")) : 544;
    return v > 0 ? v : 0x7fffffff;
}

template <int NH>
static int launch_rec_nh(int use_lstm, const float *whh, const float *bhn, const float *P, const float *h0,
                         const float *c0, float *Hout, int B, int L, int reverse_out, hipStream_t s)
{
    const dim3 grid((B + 1) / 2), block(NH * 4);
    if constexpr (NH <= 128) {
        if (use_lstm && B >= rec4_min_batch()) {      // four columns per workgroup (bit-identical to the two-column kernel)
            hipLaunchKernelGGL((lstm_rec4_kernel<NH>), dim3((B + 3) / 4), block, 0, s, (const f32x4 *)whh, P, h0, c0, Hout, B, L, reverse_out);
            CSA_HIP_CHECK(hipGetLastError());
            return CSA_OK;
        }
    }
    if (use_lstm) {
        constexpr int NL4 = NH > 128 ? 2 : 0;
        constexpr size_t shm = (size_t)4 * NL4 * NH * 4 * sizeof(f32x4);
        auto kern = lstm_rec2_kernel<NH, false, NL4>;
        if (shm > 0) {
            CSA_SET_DYN_LDS_ONCE(kern, shm);
        }
        hipLaunchKernelGGL(kern, grid, block, shm, s, (const f32x4 *)whh, (float *)P,
                           h0, c0, Hout, B, L, reverse_out, (float *)nullptr, (float *)nullptr);
    }
    else   // GRU: 3*nh/4 weights per lane (108 at nh = 144) fit the 168-VGPR budget of a 9-wave workgroup as they are
        hipLaunchKernelGGL((rec_kernel<NH, 3>), grid, block, 0, s, (const f32x4 *)whh, bhn, P, h0, c0, Hout, B, L,
                           reverse_out);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

int launch_rec_train(int nh, const float *whh_packed, float *P, const float *h0, const float *c0, float *Hout,
                     int B, int L, int reverse_out, float *Hseq, float *Cseq, hipStream_t s)
{
    const dim3 grid((B + 1) / 2), block(nh * 4);
    switch (nh) {
    case 64:  hipLaunchKernelGGL((lstm_rec2_kernel<64, true>), grid, block, 0, s, (const f32x4 *)whh_packed, P, h0, c0, Hout, B, L, reverse_out, Hseq, Cseq); break;
    case 96:  hipLaunchKernelGGL((lstm_rec2_kernel<96, true>), grid, block, 0, s, (const f32x4 *)whh_packed, P, h0, c0, Hout, B, L, reverse_out, Hseq, Cseq); break;
    case 128: hipLaunchKernelGGL((lstm_rec2_kernel<128, true>), grid, block, 0, s, (const f32x4 *)whh_packed, P, h0, c0, Hout, B, L, reverse_out, Hseq, Cseq); break;
    case 144: {
        constexpr int NL4 = 3;   // the TRAIN variant carries more live state: one more float4 per slot in LDS
        constexpr size_t shm = (size_t)4 * NL4 * 144 * 4 * sizeof(f32x4);
        auto kern = lstm_rec2_kernel<144, true, NL4>;
        CSA_SET_DYN_LDS_ONCE(kern, shm);
        hipLaunchKernelGGL(kern, grid, block, shm, s, (const f32x4 *)whh_packed, P, h0, c0, Hout, B, L, reverse_out, Hseq, Cseq);
        break;
    }
    default:
        csa_set_error_msg("rec(train): hidden size not supported (64, 96, 128, 144)");
        return CSA_ERR_UNSUPPORTED;
    }
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// GRU training forward: gates saved in place over P, h sequence into Hseq (L+1 slots).  whh_packed: gru2_pack_weights layout.
int launch_rec_train_gru(int nh, const float *whh_packed, const float *bhn, float *P, const float *h0, float *Hout, int B, int L,
                         int reverse_out, float *Hseq, hipStream_t s)
{
    const dim3 grid((B + 1) / 2), block(nh * 4);
    switch (nh) {
    case 64:  hipLaunchKernelGGL((gru_rec2_kernel<64, true>), grid, block, 0, s, (const f32x4 *)whh_packed, bhn, P, h0, Hout, B, L, reverse_out, P, Hseq); break;
    case 96:  hipLaunchKernelGGL((gru_rec2_kernel<96, true>), grid, block, 0, s, (const f32x4 *)whh_packed, bhn, P, h0, Hout, B, L, reverse_out, P, Hseq); break;
    case 128: hipLaunchKernelGGL((gru_rec2_kernel<128, true>), grid, block, 0, s, (const f32x4 *)whh_packed, bhn, P, h0, Hout, B, L, reverse_out, P, Hseq); break;
    default:
        csa_set_error_msg("rec(train, GRU): hidden size not supported (64, 96, 128)");
        return CSA_ERR_UNSUPPORTED;
    }
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
    case 144: return launch_rec_nh<144>(use_lstm, whh_packed, bhn, P, h0, c0, Hout, B, L, reverse_out, s);
    default:
        csa_set_error_msg("rec: hidden size not supported by the register-stationary kernel (64, 96, 128, 144)");
        return CSA_ERR_UNSUPPORTED;
    }
}
