// fused.hip -- the whole LSTM layer in ONE launch on BOTH fp32 pipes of the CU.
//
// nn.LSTM over the 60 levels = input projection W_ih x_t (+b) for every level  +  60 dependent cell
// steps with W_hh h_{t-1} (rnn/models/models.py:493,536).  rec.hip runs the second part on the packed-FMA
// vector pipe with W_hh stationary in registers; the MFMA pipe of those CUs idles meanwhile, and the first
// part ran as a separate GEMM launch that round-trips 47 MB of pre-activations through HBM.
//
// Here one workgroup = 12 waves, three per SIMD:
//   waves 0..7  (VALU role)  the recurrence of rec.hip, two columns per workgroup, unchanged arithmetic;
//   waves 8..11 (MFMA role)  the input projection of the SAME two columns, eight levels (16 rows) at a
//                            time, one 16x128 output slab per wave with v_mfma_f32_16x16x4_f32 (exact
//                            fp32), W_ih streamed from L2 in MFMA-operand order (host-packed, coalesced
//                            16-byte loads, one tile = 9 loads prefetched a step ahead).
// The two roles meet in LDS: the MFMA waves write chunk q+1 of the pre-activations into one half of a
// 2 x 34 KB ring while the VALU waves consume chunk q from the other half.  The only synchronisation is
// the per-step workgroup barrier the recurrence needs anyway (every wave executes exactly L + 2
// barriers); within a step the MFMA role does one 16x16 tile (36 MFMAs = 1,152 matrix-pipe cycles),
// well inside the ~2,000-cycle step of the VALU role.  No pre-activation ever reaches HBM and both
// GEMM launches of the unfused path disappear.
//
// MEASURED AFTERWARDS (tools/coexec_bench.hip): on gfx950 v_mfma_f32_* and v_pk_fma_f32 from different waves of a SIMD do
// NOT run concurrently -- together they take the sum of their separate times (they share the fp32 FMA lanes), so
// the premise "the MFMA pipe idles meanwhile" is false for fp32 and this kernel cannot beat GEMM + rec.hip.
// Kept as a parity-tested record (csa_set_fused), off by default.
//
// STATUS (round 1): parity-green (tests/test_gpu_parity.py::test_fused_and_unfused_paths_agree) but
// NOT the default: 99-109 us per layer against 34 + 56 us for GEMM + recurrent kernel.  With 16-row
// chunks every workgroup re-streams all of W_ih (288 KB) from L2 eight times per launch = 442 MB per
// layer; the MFMA role alone runs at the resulting L2 rate (75 us = 5.9 TB/s, tools/fused_bench.hip
// with -DFZ_EXP_NO_VALU), not at its 1,152-cycle matrix-pipe time.  The fix is more rows per weight
// pass (32-56 row chunks with the pre-activations in an L2-resident scratch instead of LDS), see
// DESIGN.md section 4.5.
#include "common.h"

#define PK_FMA_LO(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(h))
#define PK_FMA_HI(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(h))
#define WG_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ float fdpp_xor1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float fdpp_xor2(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

constexpr int FZ_CHUNK = 8;      // levels per projection chunk (x 2 columns = one 16-row MFMA tile)
constexpr int FZ_PROW = 512 + 32;   // LDS row stride of the pre-activation ring (floats): col1 = +32 banks

template <int NH, int KIN>
__global__ __launch_bounds__(NH * 6, 3) void fused_lstm_kernel(
    const f32x4 *__restrict__ Whh4,     // register-stationary W_hh (rec.hip / lstm2_pack_weights order)
    const f32x4 *__restrict__ Wih4,     // W_ih in MFMA-operand order (fused_pack_wih)
    const float *__restrict__ bias,     // (4*NH) permuted rows [i,g~,f,o] per unit
    const float *__restrict__ X,        // (L, B, KIN) layer input, SEQUENCE order of this layer
    const float *__restrict__ h0, const float *__restrict__ c0, float *__restrict__ Hout,
    int B, int L, int reverse_out)
{
    static_assert(NH == 128, "fused kernel is laid out for nh = 128 (8 VALU waves + 4 MFMA waves)");
    static_assert(KIN % 16 == 0, "K must be a multiple of 16");
    constexpr int KC = NH / 4, CH = 2 * KC + 4, CPY = 4 * CH;
    constexpr int XLD = KIN + 1;
    constexpr int NK4 = KIN / 16;           // 16-byte weight loads per tile
    __shared__ __attribute__((aligned(16))) float hbuf[2][2 * CPY];
    __shared__ __attribute__((aligned(16))) float pring[2][2 * FZ_CHUNK * FZ_PROW];
    __shared__ float xs[2][2 * FZ_CHUNK * XLD];

    const int tid = threadIdx.x;
    const int b0 = 2 * blockIdx.x;
    const int nchunk = (L + FZ_CHUNK - 1) / FZ_CHUNK;

    if (tid < NH * 4) {
        // ================================ VALU role: the recurrence ================================
        const int u = tid >> 2, p = tid & 3, col = p & 1, grp = p >> 1;
        int b = b0 + col;
        const bool valid = b < B;
        if (!valid) b = B - 1;
        const bool owner = grp == 1;
        f32x2 w[4][KC / 2];
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            const f32x4 v = Whh4[(size_t)i * (NH * 4) + tid];
            const int s = (4 * i) / KC, kk = (4 * i) % KC;
            w[s][kk / 2] = f32x2{v.x, v.y};
            w[s][kk / 2 + 1] = f32x2{v.z, v.w};
        }
        const float k1 = grp ? -1.44269504088896341f : -2.88539008177792681f;
        const float nb1 = grp ? 0.0f : 1.0f;
        float h = h0[(size_t)b * NH + u];
        float c = c0[(size_t)b * NH + u];
        const int slotN = 2 * u + col + 4 * (u / KC);
        const int slotS = CPY + 2 * u + (1 - col) + 4 * (u / KC);
        if (owner) { hbuf[0][slotN] = h; hbuf[0][slotS] = h; }
        // pin the loop-carried state as "produced here": otherwise hipcc waits for the c0 load with a
        // vmcnt(0) INSIDE the step (first use of c), which also waits for the previous step's h store
        asm volatile("" : "+v"(c), "+v"(h));
        const int rdoff = col * CPY + p * CH;
        const int poff = col * FZ_PROW + u * 4 + grp * 2;
        WG_BARRIER();     // P1: x chunk 0 staged by the MFMA role
        WG_BARRIER();     // P2: pre-activation chunk 0 ready
        for (int t = 0; t < L; ++t) {
#ifdef FZ_EXP_NO_VALU
            LDS_BARRIER();
            continue;
#endif
            const f32x4 *hp = (const f32x4 *)&hbuf[t & 1][rdoff];
            f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int j = 0; j < KC / 2; ++j) {
                const f32x4 hv = hp[j];
                const f32x2 ha = {hv.x, hv.y}, hb = {hv.z, hv.w};
#pragma unroll
                for (int s = 0; s < 4; ++s) PK_FMA_LO(acc[s], w[s][j], ha);
#pragma unroll
                for (int s = 0; s < 4; ++s) PK_FMA_HI(acc[s], w[s][j], hb);
            }
            const f32x2 pre = *(const f32x2 *)&pring[(t / FZ_CHUNK) & 1][(t % FZ_CHUNK) * 2 * FZ_PROW + poff];
            float r[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) r[s] = acc[s].x + fdpp_xor1(acc[s].y);
            const float v0 = r[0] + fdpp_xor2(r[2]) + pre.x;
            const float v1 = r[1] + fdpp_xor2(r[3]) + pre.y;
            const float g0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v0));
            const float t1 = fminf(__builtin_amdgcn_exp2f(k1 * v1), 1e30f);
            const float g1 = (1.0f - nb1 * t1) * __builtin_amdgcn_rcpf(1.0f + t1);
            const float ig = fdpp_xor2(g0 * g1);
            c = g0 * c + ig;
            const float tc = fminf(__builtin_amdgcn_exp2f(-2.88539008177792681f * c), 1e30f);
            h = g1 * ((1.0f - tc) * __builtin_amdgcn_rcpf(1.0f + tc));
            if (owner) {
                hbuf[(t & 1) ^ 1][slotN] = h;
                hbuf[(t & 1) ^ 1][slotS] = h;
                if (valid) {
                    // asm store: hipcc otherwise parks an s_waitcnt vmcnt(0) between this store and the
                    // step barrier (+500 cycles per step); nothing in this kernel reads Hout back
                    float *hp_out = Hout + ((size_t)(reverse_out ? L - 1 - t : t) * B + b) * NH + u;
                    asm volatile("global_store_dword %0, %1, off" : : "v"(hp_out), "v"(h) : "memory");
                }
            }
            LDS_BARRIER();
        }
    } else {
        // ================================ MFMA role: the input projection ================================
        const int mt = tid - NH * 4, lane = mt & 63, mw = mt >> 6;      // 4 waves, wave mw owns columns [128 mw, +128)
#ifdef FZ_EXP_SETPRIO
        __builtin_amdgcn_s_setprio(FZ_EXP_SETPRIO);   // diagnostic: let the (younger) MFMA waves issue ahead of the VALU waves
#endif
        const int ai = lane & 15, akq = lane >> 4;
        // stage one chunk of layer input rows: row i <-> (level t0 + i/2, column b0 + (i&1))
        auto stage_x = [&](int q) {
            const int t0 = q * FZ_CHUNK;
            float *dst = xs[q & 1];
            for (int idx = mt; idx < 2 * FZ_CHUNK * (KIN / 4); idx += 256) {
                const int i = idx / (KIN / 4), k4 = idx - i * (KIN / 4);
                const int tt = t0 + (i >> 1);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (tt < L) {
                    const int bb = min(b0 + (i & 1), B - 1);
                    v = *(const f32x4 *)(X + ((size_t)tt * B + bb) * KIN + 4 * k4);
                }
                float *d = dst + i * XLD + 4 * k4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        };
        // Weight tiles are fetched with asm loads a whole step before use and consumed behind an
        // explicit vmcnt(0) that is tied to the registers ("+v"): left to itself hipcc sinks the
        // prefetch next to the MFMAs (vmcnt(1) after every load) and the role becomes L2-latency bound
        // (2,800 cycles per tile measured instead of 1,152).  Two register sets alternate (no moves).
        f32x4 wA[NK4], wB[NK4];
        static_assert(NK4 == 8 || NK4 == 9, "weight tile = 8 or 9 sixteen-byte loads");
#define FZ_WLOAD(W, TILE)                                                                          \
    {                                                                                              \
        const f32x4 *wp_ = Wih4 + ((size_t)(mw * 8 + (TILE)) * NK4) * 64 + lane;                   \
        _Pragma("unroll") for (int k4 = 0; k4 < NK4; ++k4)                                         \
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(W[k4]) : "v"(wp_ + (size_t)k4 * 64) : "memory"); \
    }
#define FZ_TIE8(W) "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]), "+v"(W[4]), "+v"(W[5]), "+v"(W[6]), "+v"(W[7])
#define FZ_WAIT(W)                                                                                 \
    {                                                                                              \
        if constexpr (NK4 == 9) asm volatile("s_waitcnt vmcnt(0)" : FZ_TIE8(W), "+v"(W[NK4 - 1]) : : "memory"); \
        else asm volatile("s_waitcnt vmcnt(0)" : FZ_TIE8(W) : : "memory");                         \
    }
#define FZ_BARRIER_W(W)                                                                            \
    {                                                                                              \
        if constexpr (NK4 == 9)                                                                    \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : FZ_TIE8(W), "+v"(W[NK4 - 1]) : : "memory"); \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : FZ_TIE8(W) : : "memory"); \
    }
        // one 16x16 output tile (columns [128 mw + 16 tile, +16)) of chunk q from xs[q&1] into pring[q&1]
        auto do_tile = [&](int q, int tile, const f32x4 *wt) {
            const float *xa = xs[q & 1] + ai * XLD + akq;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k4 = 0; k4 < NK4; ++k4) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[16 * k4 + 0], wt[k4].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[16 * k4 + 4], wt[k4].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[16 * k4 + 8], wt[k4].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[16 * k4 + 12], wt[k4].w, acc1, 0, 0, 0);
            }
            const int n = mw * 128 + tile * 16 + ai;
            const float bv = bias[n];
            float *pr = pring[q & 1] + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[(akq * 4 + r) * FZ_PROW] = (acc0[r] + acc1[r]) + bv;
        };
        stage_x(0);
        FZ_WLOAD(wA, 0)
        FZ_BARRIER_W(wA)      // P1 (also: x chunk 0 visible to all four MFMA waves)
        for (int tile = 0; tile < 8; tile += 2) {      // chunk 0 at full speed (overlaps the VALU role's weight load)
            FZ_WLOAD(wB, tile + 1)
            do_tile(0, tile, wA);
            FZ_WAIT(wB)
            FZ_WLOAD(wA, (tile + 2) & 7)               // after the last pair: tile 0 of chunk 1
            do_tile(0, tile + 1, wB);
            FZ_WAIT(wA)
        }
        if (nchunk > 1) stage_x(1);
        FZ_BARRIER_W(wA)      // P2
        // main loop: step t produces tile t%8 of chunk t/8 + 1; even steps compute from wA and prefetch wB
#define FZ_MFMA_STEP(T, WC, WN)                                                                    \
    {                                                                                              \
        const int q_ = (T) / FZ_CHUNK + 1, tile_ = (T) % FZ_CHUNK;                                 \
        if (q_ < nchunk) {                                                                         \
            FZ_WLOAD(WN, (tile_ + 1) & 7)                                                          \
            do_tile(q_, tile_, WC);                                                                \
            if (tile_ == FZ_CHUNK - 1 && q_ + 1 < nchunk) stage_x(q_ + 1);                         \
        }                                                                                          \
        FZ_BARRIER_W(WN)                                                                           \
    }
        for (int t = 0; t < L; t += 2) {
            FZ_MFMA_STEP(t, wA, wB)
            if (t + 1 < L) FZ_MFMA_STEP(t + 1, wB, wA)
        }
#undef FZ_MFMA_STEP
#undef FZ_WLOAD
#undef FZ_WAIT
#undef FZ_BARRIER_W
#undef FZ_TIE8
    }
}

// W_ih (4*nh rows permuted [n'], K) -> MFMA-operand order: [wave 4][tile 8][K/16][lane 64][4]
// element e of that float4 is the B operand of k-step 4*k4+e: W[n' = 128 w + 16 tile + (lane&15)][k = 16 k4 + 4 e + (lane>>4)]
size_t fused_packed_floats(int nh, int K) { return (size_t)4 * nh * K; }
void fused_pack_wih(int nh, int K, const float *w_perm, float *packed)
{
    const int nk4 = K / 16;
    for (int w = 0; w < 4; ++w)
        for (int tile = 0; tile < 8; ++tile)
            for (int k4 = 0; k4 < nk4; ++k4)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int n = w * (nh) + tile * 16 + (lane & 15);
                        const int k = 16 * k4 + 4 * e + (lane >> 4);
                        packed[((((size_t)(w * 8 + tile) * nk4 + k4) * 64 + lane) * 4) + e] = w_perm[(size_t)n * K + k];
                    }
}

int launch_fused_lstm(int nh, int K, const float *whh_packed, const float *wih_packed, const float *bias,
                      const float *X, const float *h0, const float *c0, float *Hout, int B, int L, int reverse_out,
                      hipStream_t s)
{
    if (nh != 128 || (K != 128 && K != 144)) {
        csa_set_error_msg("fused_lstm: built for nh = 128 and K in {128, 144}");
        return CSA_ERR_UNSUPPORTED;
    }
    const dim3 grid((B + 1) / 2), block(nh * 6);
    if (K == 144)
        hipLaunchKernelGGL((fused_lstm_kernel<128, 144>), grid, block, 0, s, (const f32x4 *)whh_packed,
                           (const f32x4 *)wih_packed, bias, X, h0, c0, Hout, B, L, reverse_out);
    else
        hipLaunchKernelGGL((fused_lstm_kernel<128, 128>), grid, block, 0, s, (const f32x4 *)whh_packed,
                           (const f32x4 *)wih_packed, bias, X, h0, c0, Hout, B, L, reverse_out);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
