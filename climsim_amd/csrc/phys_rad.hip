// phys_rad.hip -- the physical radiation scheme of the physRNN "Hidden" radiation graphs, as
// serialised in rnn/saved_models/physRNN-Hidden_*_num4050_BEST_script_cpu.pt (`radiative_transfer`; the current source
// splits it differently, rnn/models/models_phys.py:1272-1584).  Helpers restated from rnn/models/physics_rad.py:
//   :34 interpolate_tlev_batchlast   :51 outgoing_lw   :60 reftrans_lw   :96 lw_solver_noscat_batchlast
//   :139 calc_ref_trans_sw           :332 adding_ica_sw_inference
// and rnn/layers.py gasopt_mlp (LW gas optics 18 -> 64 -> 64 -> 256, reduced to 16 g-points by two Linear(128, 16)).
//
// After phys_decode_kernel<.., RAD> has written the MLP inputs (XG, XR), the per-level scalars (RS: dry-air column, updated
// temperature) and the cloud optical depth (CL), one call runs two kernels:
//   rad_optics_kernel     the three MLPs on the matrix pipe, 32 (level, column) rows per wave, activations never leaving the CU
//                         (gas optics 24 -> 64 -> 64 -> 256 -> k-distribution reduction to 16 + 16; SW head 24 -> 32 -> 48)
//   phys_rad_solve_kernel one workgroup per grid column; two-stream coefficients for its 60 x 16 (level, g-point) cells in
//                         parallel, then the level recurrences (LW no-scattering sweep on wave 0, SW adding method on wave 1,
//                         one lane per g-point), flux sums, heating rate added onto out_lev[:, :, 0], six surface fluxes.
// The LW downward source equals the upward source, as in the serialised graph.
#include "phys.h"

template <int N> __device__ __forceinline__ float pr_sum(float v)
{
#pragma unroll
    for (int o = N / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int N> __device__ __forceinline__ float pr_max(float v)
{
#pragma unroll
    for (int o = N / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float pr_pow8(float x) { x *= x; x *= x; return x * x; }
__device__ __forceinline__ float pr_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float pr_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// ------------------------------------------------------------------------------------------------------------------
// Fused optics: the three MLPs of the scheme for a tile of 32 (level, column) rows per wave, activations never leaving the
// CU.  Per wave: XG (32 x 24) -> Linear 64 + Softsign -> Linear 64 + Softsign -> Linear 256 in eight 32-column tiles, each tile
// turned into k-point optical depths N_dry (s x + m)^8 / Planck-fraction logits x^2 in the accumulator registers, passed through
// LDS and contracted at once with its 32-row slice of the two k-distribution reductions (128 -> 16 each, one 32 x 32
// accumulator holding both: columns 0-15 optical depth, 16-31 Planck fraction); then XR (32 x 24) -> Linear 32 + Softsign ->
// Linear 48.  All products on v_mfma_f32_32x32x2_f32 (exact fp32): A operand = activations from LDS (lane = row), B operand =
// weights straight from global memory / L2 as float4 -- lanes 0-31 take k..k+3 and lanes 32-63 k+4..k+7 of weight row n, so
// MFMA e of a group contracts the pair (k+e, k+4+e) on both operands.  516 MFMAs per tile; one wave (one tile) per workgroup.
// Replaced seven GEMM launches + one elementwise kernel: 99 -> 27 us at 384 columns (profiles/r2_physrnn_rad_384_*).
#define RO_LD 68            // multiples of 4 floats: the A operand is read as one ds_read_b128 per MFMA group
#define RO_LX 28
struct RadOptics {
    const float *XG, *XR, *RS;
    const float *w1, *b1, *w2, *b2, *w3, *b3, *r1w, *r1b, *r2w, *r2b, *s1w, *s1b, *s2w, *s2b, *ystd, *ymean;
    float *TP, *S2;
    int M;
};

// weights of one 32-column tile, K deep: G = K / 8 float4 per lane (lane's weight row, k-quad 4 * half of every group of 8)
template <int G> struct RoW { f32x4 v[G]; };
template <int G> __device__ __forceinline__ RoW<G> ro_load(const float *__restrict__ w_row)
{
    RoW<G> w;
#pragma unroll
    for (int g = 0; g < G; ++g) w.v[g] = *(const f32x4 *)(w_row + 8 * g);
    return w;
}
// A operand held in registers (the same activations feed every 32-column tile of a layer)
template <int G> __device__ __forceinline__ RoW<G> ro_loada(const float *act_row)
{
    RoW<G> x;
#pragma unroll
    for (int g = 0; g < G; ++g) x.v[g] = *(const f32x4 *)(act_row + 8 * g);
    return x;
}
template <int G> __device__ __forceinline__ f32x16 ro_mma_r(f32x16 acc, const RoW<G> &x, const RoW<G> &w)
{
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.v[g][e], w.v[g][e], acc, 0, 0, 0);
    return acc;
}
// live == false: this lane's weight row is all zeros (the select sits here, at the point of use: next to the load it would
// make the wave wait for the prefetch it has just issued)
template <int G> __device__ __forceinline__ f32x16 ro_mma(f32x16 acc, const float *act_row /* LDS: lane's row + 4 * half */, const RoW<G> &w,
                                                         bool live = true)
{
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const f32x4 x = *(const f32x4 *)(act_row + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[e], live ? w.v[g][e] : 0.0f, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ f32x16 ro_zero()
{
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}
// v / (1 + |v|) through v_rcp_f32 (1 ulp; an IEEE division is ~10 instructions, and the optics kernels spend a third of their wave
// time issuing vector instructions: profiles/r2_physrnn_e3sm_384_sq_pmc.json)
__device__ __forceinline__ float ro_softsign(float v) { return v * __builtin_amdgcn_rcpf(1.0f + fabsf(v)); }

// One wave per workgroup and all LDS traffic wave-private: LDS executes one wave's instructions in order, so a write followed by
// a read of other lanes' data needs no s_barrier, only a compiler fence (ro_fence).  Every tile's weights are requested one or two
// tiles ahead of their use (L2 round trips), and in the 256-wide layer the VALU epilogue of tile t ((s x + m)^8 N_dry, LDS
// transpose) is issued between the MFMAs of tile t + 1 -- an in-order wave would otherwise idle the matrix pipe for a third of
// the time.
__device__ __forceinline__ void ro_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }

#define RO_WAVES 1          // independent waves per workgroup (1 / 4 measured: 37 / 46 us at 384 columns -- four waves need 119 KB of LDS,
                            // one workgroup per CU, 180 CUs busy)
__global__ __launch_bounds__(64 * RO_WAVES) void rad_optics_kernel(RadOptics a)
{
    __shared__ __attribute__((aligned(16))) float sA[RO_WAVES][32 * RO_LD];      // H1, then Y of even tiles   (SW: S1)
    __shared__ __attribute__((aligned(16))) float sB[RO_WAVES][32 * RO_LD];      // H2
    __shared__ __attribute__((aligned(16))) float sC[RO_WAVES][32 * RO_LD];      // Y of odd tiles
    __shared__ __attribute__((aligned(16))) float sX[RO_WAVES][32 * RO_LX];      // XG                         (SW: XR)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
    const int row0 = ((int)blockIdx.x * RO_WAVES + wave) * 32, M = a.M;
    if (row0 >= M) return;                           // (no workgroup barrier anywhere below)
    float *bA = sA[wave], *bB = sB[wave], *bC = sC[wave], *bX = sX[wave];
    // accumulator register i of this lane belongs to tile row (i & 3) + 8 (i >> 2) + 4 half, tile column n
    auto drow = [&](int i) { return (i & 3) + 8 * (i >> 2) + 4 * half; };
    constexpr int G1 = PH_XG_K / 8, GS = PH_XR_K / 8;
    static_assert(PH_XG_K == 24 && PH_XR_K == 24, "tile loader");
    // an input tile = 32 consecutive rows = 768 consecutive floats: three float4 per lane, all requested before any is used
    // (a rolled loop of dependent load -> LDS store pairs cost ~1.5 us per trip: 20 us of this kernel's first version)
    auto load_tile = [&](const float *__restrict__ X) {
        f32x4 v[3];
        const size_t tile0 = (size_t)row0 * 24, last4 = (size_t)M * 24 - 4;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const size_t e = tile0 + 4 * (lane + 64 * j);
            v[j] = *(const f32x4 *)(X + (e < last4 ? e : last4));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = 4 * (lane + 64 * j), rr = e / 24, k = e - rr * 24;
            *(f32x4 *)(bX + rr * RO_LX + k) = v[j];
        }
    };

    if (blockIdx.y == 1) {
        // ---- SW optical-property head: 24 -> 32 (Softsign) -> 48, its own workgroups (off the gas-optics critical path) ----
        const RoW<GS> ws1 = ro_load<GS>(a.s1w + (size_t)n * PH_XR_K + 4 * half);
        const RoW<4> ws2a = ro_load<4>(a.s2w + (size_t)n * 32 + 4 * half), ws2b = ro_load<4>(a.s2w + (size_t)(32 + (n & 15)) * 32 + 4 * half);
        load_tile(a.XR);
        ro_fence();
        {
            const f32x16 acc = ro_mma<GS>(ro_zero(), bX + n * RO_LX + 4 * half, ws1);
            const float b = a.s1b[n];
#pragma unroll
            for (int i = 0; i < 16; ++i) bA[drow(i) * RO_LD + n] = ro_softsign(acc[i] + b);
        }
        ro_fence();
        const RoW<4> x = ro_loada<4>(bA + n * RO_LD + 4 * half);
        f32x16 acc = ro_mma_r<4>(ro_zero(), x, ws2a);
        float b = a.s2b[n];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = row0 + drow(i);
            if (row < M) a.S2[(size_t)row * 48 + n] = acc[i] + b;
        }
        acc = ro_mma<4>(ro_zero(), bA + n * RO_LD + 4 * half, ws2b, n < 16);
        b = n < 16 ? a.s2b[32 + n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = row0 + drow(i);
            if (n < 16 && row < M) a.S2[(size_t)row * 48 + 32 + n] = acc[i] + b;
        }
        return;
    }

    // ---- LW gas optics ----
    RoW<G1> w1a = ro_load<G1>(a.w1 + (size_t)n * PH_XG_K + 4 * half), w1b = ro_load<G1>(a.w1 + (size_t)(32 + n) * PH_XG_K + 4 * half);
    load_tile(a.XG);
    float cd[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cd[i] = a.RS[(size_t)min(row0 + drow(i), M - 1) * 2];
    RoW<8> wn = ro_load<8>(a.w2 + (size_t)n * 64 + 4 * half);                       // layer 2, tile 0
    RoW<8> wm = ro_load<8>(a.w2 + (size_t)(32 + n) * 64 + 4 * half);                // layer 2, tile 1
    ro_fence();
    // layer 1: 24 -> 64
    {
        const RoW<G1> x = ro_loada<G1>(bX + n * RO_LX + 4 * half);
        f32x16 acc = ro_mma_r<G1>(ro_zero(), x, w1a);
        float b = a.b1[n];
#pragma unroll
        for (int i = 0; i < 16; ++i) bA[drow(i) * RO_LD + n] = ro_softsign(acc[i] + b);
        acc = ro_mma_r<G1>(ro_zero(), x, w1b);
        b = a.b1[32 + n];
#pragma unroll
        for (int i = 0; i < 16; ++i) bA[drow(i) * RO_LD + 32 + n] = ro_softsign(acc[i] + b);
    }
    ro_fence();
    // layer 2: 64 -> 64
    {
        const RoW<8> x = ro_loada<8>(bA + n * RO_LD + 4 * half);
        f32x16 acc = ro_mma_r<8>(ro_zero(), x, wn);
        wn = ro_load<8>(a.w3 + (size_t)n * 64 + 4 * half);                          // layer 3, tile 0
        float b = a.b2[n];
#pragma unroll
        for (int i = 0; i < 16; ++i) bB[drow(i) * RO_LD + n] = ro_softsign(acc[i] + b);
        acc = ro_mma_r<8>(ro_zero(), x, wm);
        wm = ro_load<8>(a.w3 + (size_t)(32 + n) * 64 + 4 * half);                   // layer 3, tile 1
        b = a.b2[32 + n];
#pragma unroll
        for (int i = 0; i < 16; ++i) bB[drow(i) * RO_LD + 32 + n] = ro_softsign(acc[i] + b);
    }
    ro_fence();
    // layer 3 (64 -> 256) tile by tile, each tile reduced 32 -> (16 | 16) at once.  Iteration t: the MFMAs of tile t + 1 with the
    // VALU epilogue of tile t issued in between, then the reduction of tile t; the weights of tile t + 2 and the reduction slice
    // and constants of tile t + 1 are in flight meanwhile.  (Fully unrolled 36.6 us, rolled 40.9 us per 384-column launch.)
    const RoW<8> hx = ro_loada<8>(bB + n * RO_LD + 4 * half);      // H2: the A operand of all eight tiles
    auto red_w = [&](int t) {                        // the 32-deep slice of the (16 | 16)-column reduction that tile t feeds
        return ro_load<4>((t < 4 ? a.r1w : a.r2w) + (size_t)(n & 15) * 128 + (t & 3) * 32 + 4 * half);
    };
    RoW<4> wrn = red_w(0);
    float bn = a.b3[n], sdn = a.ystd[n], mnn = a.ymean[n];
    f32x16 tp = ro_zero();
    f32x16 accC = ro_mma_r<8>(ro_zero(), hx, wn);                                   // tile 0
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const RoW<4> wr = wrn;
        const float b = bn, sd = sdn, mn = mnn;
        if (t + 2 < 8) wn = ro_load<8>(a.w3 + (size_t)((t + 2) * 32 + n) * 64 + 4 * half);
        if (t + 1 < 8) {
            wrn = red_w(t + 1);
            bn = a.b3[(t + 1) * 32 + n]; sdn = a.ystd[((t + 1) & 3) * 32 + n]; mnn = a.ymean[((t + 1) & 3) * 32 + n];
        }
        float *Y = (t & 1) ? bC : bA;
        const bool tau = t < 4;                      // optical-depth half of the layer: N_dry (s x + m)^8; else squared logit
        f32x16 accN = ro_zero(), accM = ro_zero();   // two accumulation chains (even / odd MFMAs)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (t < 7) {
                    if (e & 1) accM = __builtin_amdgcn_mfma_f32_32x32x2f32(hx.v[g][e], wm.v[g][e], accM, 0, 0, 0);
                    else accN = __builtin_amdgcn_mfma_f32_32x32x2f32(hx.v[g][e], wm.v[g][e], accN, 0, 0, 0);
                }
                if (!(e & 1)) {
                    const int i = 2 * g + (e >> 1);  // accumulator register i of tile t -> LDS, transposed for the reduction
                    const float v = accC[i] + b;
                    Y[drow(i) * RO_LD + n] = tau ? cd[i] * pr_pow8(sd * v + mn) : v * v;
                }
                __builtin_amdgcn_sched_barrier(0x3F4);   // memory and scalar instructions may move; MFMA and VALU keep this order
            }
        }
        ro_fence();
        tp = ro_mma<4>(tp, Y + n * RO_LD + 4 * half, wr, tau ? n < 16 : n >= 16);
#pragma unroll
        for (int i = 0; i < 16; ++i) accC[i] = accN[i] + accM[i];
        wm = wn;
    }
    const float b = n < 16 ? a.r1b[n] : a.r2b[n - 16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = row0 + drow(i);
        if (row < M) a.TP[(size_t)row * 32 + n] = tp[i] + b;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// physics_rad_e3sm generation: SW optical properties of every (level, column) row, on the matrix pipe like rad_optics_kernel.
// A wave owns a tile of 32 rows; for each humidity variant (the two largest regions) and each gas-optics model (absorption,
// Rayleigh) it runs 8 -> 32 -> 32 -> 128 (112 used) tile by tile, turns every 32-column tile into k-point optical depths
// col_dry (ystd y + ymean)^8 in the accumulator registers, passes it through LDS and contracts it at once with its slice of
// the model's 112 -> 16 reduction -- one 32 x 32 accumulator for everything: columns 0-15 absorption, 16-31 Rayleigh, both
// variants summed (the average of the two is taken after the linear reduction).  592 v_mfma_f32_32x32x2_f32 per tile.
// The weights (SWG_FLOATS, 69 KB, rows padded by 4 floats against bank conflicts of the B-operand reads) sit in LDS, shared by
// the four waves of a workgroup; everything else is wave-private.
// S2 row: [tau_sw (16) | ssa (16) | asymmetry (16)] -- final values (the solver applies no activation for this generation).
// First version (fp32 VALU, two lanes per row): 66.8 us per 384-column call; this one: see profiles/r2_physrnn_e3sm_*.
#define SG_WAVES 4
#define SG_LD 36            // 32-wide activations: row stride 32 + 4 floats (b128 operand reads, as RO_LD)
__global__ __launch_bounds__(64 * SG_WAVES) void rad_sw_gas_kernel(const float *__restrict__ XR, const float *__restrict__ swg, const float *__restrict__ CS,
                                                                   float *__restrict__ S2, int M, int B, int ilev)
{
    __shared__ __attribute__((aligned(16))) float sw[SWG_FLOATS];
    __shared__ __attribute__((aligned(16))) float sA[SG_WAVES][2][32 * SG_LD];   // per variant: H1, then the k-point tile Y
    __shared__ __attribute__((aligned(16))) float sB[SG_WAVES][2][32 * SG_LD];   // per variant: H2; [0] at the end: the reduced tile
    __shared__ __attribute__((aligned(16))) float sX[SG_WAVES][32 * RO_LX];      // XR rows: [x (7), x2' | col_dry (2) | ... | variant-2 input (8) at 16]
    {   // weights -> LDS: all sixteen loads of a thread in flight before the first store (a rolled load -> store loop pays one
        // L2 round trip per trip)
        constexpr int N4 = SWG_FLOATS / 4, T = 64 * SG_WAVES, J = (N4 + T - 1) / T;
        f32x4 v[J];
#pragma unroll
        for (int j = 0; j < J; ++j) v[j] = ((const f32x4 *)swg)[min((int)threadIdx.x + T * j, N4 - 1)];
#pragma unroll
        for (int j = 0; j < J; ++j) if ((int)threadIdx.x + T * j < N4) ((f32x4 *)sw)[threadIdx.x + T * j] = v[j];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
    const int row0 = ((int)blockIdx.x * SG_WAVES + wave) * 32;
    float *bX = sX[wave];
    auto drow = [&](int i) { return (i & 3) + 8 * (i >> 2) + 4 * half; };
    static_assert(PH_XR_K == 24, "tile loader");
    if (row0 < M) {
        f32x4 v[3];
        const size_t tile0 = (size_t)row0 * 24, last4 = (size_t)M * 24 - 4;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const size_t e = tile0 + 4 * (lane + 64 * j);
            v[j] = *(const f32x4 *)(XR + (e < last4 ? e : last4));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int e = 4 * (lane + 64 * j), rr = e / 24, k = e - rr * 24;
            *(f32x4 *)(bX + rr * RO_LX + k) = v[j];
        }
    }
    __syncthreads();
    if (row0 >= M) return;                           // (no workgroup barrier below)
    if (lane < 32) {                                  // second variant: the same inputs with its own humidity feature
        float *r = bX + lane * RO_LX;
        const f32x4 a = *(const f32x4 *)r, c = *(const f32x4 *)(r + 4);
        *(f32x4 *)(r + 16) = f32x4{a.x, a.y, c.w, a.w};
        *(f32x4 *)(r + 20) = f32x4{c.x, c.y, c.z, 0.0f};
    }
    ro_fence();
    // the two variants are independent until the reduction: their stages are written side by side so that the VALU epilogue
    // of one overlaps the MFMAs of the other
    float *bA0 = sA[wave][0], *bA1 = sA[wave][1], *bB0 = sB[wave][0], *bB1 = sB[wave][1];
    float cd0[16], cd1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { cd0[i] = bX[drow(i) * RO_LX + 8]; cd1[i] = bX[drow(i) * RO_LX + 9]; }
    const RoW<1> xa = ro_loada<1>(bX + n * RO_LX + 4 * half), xb = ro_loada<1>(bX + n * RO_LX + 16 + 4 * half);
    f32x16 tp0 = ro_zero(), tp1 = ro_zero();
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const float *W = sw + SWG_MODEL0 + m * SWG_MODEL_FLOATS;
        {   // layer 1: 8 -> 32 (the weight of input column 7 is zero: variant 0 leaves the other variant's feature there)
            const RoW<1> w = ro_load<1>(W + SWG_W1 + n * SWG_LD1 + 4 * half);
            const f32x16 a0 = ro_mma_r<1>(ro_zero(), xa, w), a1 = ro_mma_r<1>(ro_zero(), xb, w);
            const float b = W[SWG_B1 + n];
#pragma unroll
            for (int i = 0; i < 16; ++i) { bA0[drow(i) * SG_LD + n] = ro_softsign(a0[i] + b); bA1[drow(i) * SG_LD + n] = ro_softsign(a1[i] + b); }
        }
        ro_fence();
        {   // layer 2: 32 -> 32
            const RoW<4> x0 = ro_loada<4>(bA0 + n * SG_LD + 4 * half), x1 = ro_loada<4>(bA1 + n * SG_LD + 4 * half);
            const RoW<4> w = ro_load<4>(W + SWG_W2 + n * SWG_LDK + 4 * half);
            const f32x16 a0 = ro_mma_r<4>(ro_zero(), x0, w), a1 = ro_mma_r<4>(ro_zero(), x1, w);
            const float b = W[SWG_B2 + n];
#pragma unroll
            for (int i = 0; i < 16; ++i) { bB0[drow(i) * SG_LD + n] = ro_softsign(a0[i] + b); bB1[drow(i) * SG_LD + n] = ro_softsign(a1[i] + b); }
        }
        ro_fence();
        const RoW<4> h0 = ro_loada<4>(bB0 + n * SG_LD + 4 * half), h1 = ro_loada<4>(bB1 + n * SG_LD + 4 * half);
        const float *R = sw + SWG_RED + m * SWG_RED_FLOATS;
        const bool live = m == 0 ? n < 16 : n >= 16;
        // layer 3 tile by tile, each tile reduced at once.  Pinned issue order (as in rad_optics_kernel): the (.)^8 epilogue of one
        // variant's tile sits between the MFMAs of the other variant's tile / of the reduction, one accumulator element per MFMA;
        // the next tile's first chain is issued while the last epilogue's LDS writes land.
        RoW<4> w = ro_load<4>(W + SWG_W3 + n * SWG_LDK + 4 * half);
        f32x16 a0 = ro_mma_r<4>(ro_zero(), h0, w);
#pragma unroll
        for (int t = 0; t < SWG_NKP / 32; ++t) {
            RoW<4> wr = ro_load<4>(R + (n & 15) * SWG_LDR + t * 32 + 4 * half);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) wr.v[g][e] = live ? wr.v[g][e] : 0.0f;
            const float b = W[SWG_B3 + t * 32 + n], sd = W[SWG_YSTD + t * 32 + n], mn = W[SWG_YMEAN + t * 32 + n];
            f32x16 a1 = ro_zero();
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(h1.v[g][e], w.v[g][e], a1, 0, 0, 0);
                    const int i = 4 * g + e;
                    bA0[drow(i) * SG_LD + n] = cd0[i] * pr_pow8(sd * (a0[i] + b) + mn);
                    __builtin_amdgcn_sched_barrier(0x3F4);   // memory and scalar instructions may move; MFMA and VALU keep this order
                }
            ro_fence();
            {
                const RoW<4> x0 = ro_loada<4>(bA0 + n * SG_LD + 4 * half);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        tp0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.v[g][e], wr.v[g][e], tp0, 0, 0, 0);
                        const int i = 4 * g + e;
                        bA1[drow(i) * SG_LD + n] = cd1[i] * pr_pow8(sd * (a1[i] + b) + mn);
                        __builtin_amdgcn_sched_barrier(0x3F4);
                    }
            }
            ro_fence();
            if (t + 1 < SWG_NKP / 32) {
                w = ro_load<4>(W + SWG_W3 + ((t + 1) * 32 + n) * SWG_LDK + 4 * half);
                a0 = ro_mma_r<4>(ro_zero(), h0, w);
            }
            {
                const RoW<4> x1 = ro_loada<4>(bA1 + n * SG_LD + 4 * half);
                tp1 = ro_mma_r<4>(tp1, x1, wr);
            }
        }
        ro_fence();
    }
    float *bB = bB0;
    {   // mean of the two variants + bias; through LDS so that one lane sees absorption and Rayleigh of a (row, g-point)
        const float b = sw[SWG_RED + (n >> 4) * SWG_RED_FLOATS + 16 * SWG_LDR + (n & 15)];
#pragma unroll
        for (int i = 0; i < 16; ++i) bB[drow(i) * SG_LD + n] = (tp0[i] + tp1[i]) * 0.5f + b;
    }
    ro_fence();
    // gas + cloud -> layer optical depth, single-scattering albedo, asymmetry (region g is g-point g)
    const int g = lane & 15;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = (lane >> 4) + 4 * j, row = row0 + r;
        if (row >= M) continue;
        const float t_abs = pr_softplus(bB[r * SG_LD + g]) * 0.01f + 1e-9f, t_sca = pr_softplus(bB[r * SG_LD + 16 + g]) * 0.01f;
        const int L = row / B, b = row - L * B;
        float c_tau = 0.0f, c_sca = 0.0f, c_asy = 0.0f;
        if (L >= ilev) {
            const float *cs = CS + ((size_t)(L - ilev) * B + b) * 48;
            c_tau = cs[g]; c_sca = cs[16 + g]; c_asy = cs[32 + g];
        }
        const float tau = (t_abs + t_sca) + c_tau, sca = t_sca + c_sca;
        float *o = S2 + (size_t)row * 48;
        o[g] = tau; o[16 + g] = sca * __builtin_amdgcn_rcpf(tau); o[32 + g] = (c_asy * c_sca) * __builtin_amdgcn_rcpf(sca);
    }
}

// ---- SW gas optics of the nx21 generation (the frozen exports) -------------------------------------------------------------------
// Two models (absorption, Rayleigh) 7 -> 32 -> 32 -> ng, Softsign, tau = N_dry y^8 1e-17, each evaluated for the humidity of the two
// largest regions of the level (XR row: [T, ln p, h2o_1^(1/4), o3^(1/4), co2, n2o, ch4, h2o_2^(1/4), N_dry_1, N_dry_2], normalised by the
// decoder); per (row, g-point) ONE of the two humidity variants is taken by the coin mask_u < 0.5.  14 kFLOP per row -- 0.33 GFLOP per
// 384-column call, a fiftieth of the GRU work -- so this is a vector kernel, and what it costs is the LENGTH of a lane's dependent
// chain, not throughput (22 us at 48 columns, 24 us at 384 with one lane per (row, combination) evaluating a whole MLP: 1,792 FMAs).
// Round 3: SIXTEEN lanes per row -- lane 16 r + 4 c + q evaluates, for combination c = 2 variant + model of row r, the hidden units
// n = 4 j + q (j = 0..7) of both layers and the outputs g = 4 j + q (j = 0..3): 440 FMAs; the four q-lanes of a quad exchange their
// activations by DPP quad broadcasts, the four combinations of a row their optical depths through LDS (all sixteen lanes sit in one
// wave), and lane (c, q) finishes g-point 4 c + q (coin, clamps, gas + cloud -> tau, ssa, g).  Weights in LDS with row strides of
// 36 floats (W2, W3) so that the four adjacent rows a quad reads as float4 fall into different banks.
// Padded g-points (ng < 16) get tau 1, ssa 0, g 0: they carry no flux.
#define SX_ROWS 16
#define SX_LD 36
#define SX_W1 0                                 // per model in LDS: W1 (32, 8), b1 (32), W2 (32, 36), b2 (32), W3 (16, 36), b3 (16)
#define SX_B1 (SX_W1 + 32 * 8)
#define SX_W2 (SX_B1 + 32)
#define SX_B2 (SX_W2 + 32 * SX_LD)
#define SX_W3 (SX_B2 + 32)
#define SX_B3 (SX_W3 + 16 * SX_LD)
#define SX_MODEL (SX_B3 + 16)
template <int Q> __device__ __forceinline__ float sx_quad(float v)      // the value lane Q of this lane's quad holds
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), Q * 0x55, 0xF, 0xF, true));
}
__global__ __launch_bounds__(256) void rad_sw_gas16_kernel(const float *__restrict__ XR, const float *__restrict__ swx, const float *__restrict__ CS,
                                                          const float *__restrict__ mask_u, float *__restrict__ S2, int M, int B, int ilev, int ng, int ngk,
                                                          int npass, int mem_B, int mem_off)
{
    __shared__ __attribute__((aligned(16))) float sw[2 * SX_MODEL];
    __shared__ __attribute__((aligned(16))) float sred[2 * SWX_RED_FLOATS];
    __shared__ float st[SX_ROWS][4][16];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * SWX_MODEL_FLOATS; i += 256) {        // global block (row stride 32) -> LDS (row stride 36 for W2, W3)
        const int mdl = i / SWX_MODEL_FLOATS, o = i - mdl * SWX_MODEL_FLOATS;
        int dst;
        if (o < SWX_B1) dst = SX_W1 + o;
        else if (o < SWX_W2) dst = SX_B1 + (o - SWX_B1);
        else if (o < SWX_B2) { const int e = o - SWX_W2; dst = SX_W2 + (e >> 5) * SX_LD + (e & 31); }
        else if (o < SWX_W3) dst = SX_B2 + (o - SWX_B2);
        else if (o < SWX_B3) { const int e = o - SWX_W3; dst = SX_W3 + (e >> 5) * SX_LD + (e & 31); }
        else dst = SX_B3 + (o - SWX_B3);
        sw[mdl * SX_MODEL + dst] = swx[SWX_MODEL0 + i];
    }
    if (ngk > 0) for (int i = tid; i < 2 * SWX_RED_FLOATS; i += 256) sred[i] = swx[SWX_RED + i];
    __syncthreads();
    const int r = tid >> 4, c = (tid >> 2) & 3, q = tid & 3, variant = c >> 1, model = c & 1;
    // npass row groups per workgroup: 1 where the chain length decides (a few hundred columns), 4 at shard size, where the weight
    // block's trip to LDS per workgroup does (1.412 against 1.397 ms per 2,700-column call with one pass)
    for (int pass = 0; pass < npass; ++pass) {
    const int row = (blockIdx.x * npass + pass) * SX_ROWS + r, rowc = min(row, M - 1);
    const float *xr = XR + (size_t)rowc * PH_XR_K;
    float x[8];
#pragma unroll
    for (int k = 0; k < 7; ++k) x[k] = xr[k];
    if (variant) x[2] = xr[7];
    x[7] = 0.0f;
    const float col = xr[8 + variant];
    const float *W = sw + model * SX_MODEL;
    float h1[8], h2[8], hf[32];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = 4 * j + q;
        float a = W[SX_B1 + n];
        const f32x4 w0 = *(const f32x4 *)(W + SX_W1 + n * 8), w1 = *(const f32x4 *)(W + SX_W1 + n * 8 + 4);
        a = fmaf(x[0], w0.x, a); a = fmaf(x[1], w0.y, a); a = fmaf(x[2], w0.z, a); a = fmaf(x[3], w0.w, a);
        a = fmaf(x[4], w1.x, a); a = fmaf(x[5], w1.y, a); a = fmaf(x[6], w1.z, a);
        h1[j] = a / (fabsf(a) + 1.0f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { hf[4 * j] = sx_quad<0>(h1[j]); hf[4 * j + 1] = sx_quad<1>(h1[j]); hf[4 * j + 2] = sx_quad<2>(h1[j]); hf[4 * j + 3] = sx_quad<3>(h1[j]); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = 4 * j + q;
        float a = W[SX_B2 + n];
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
            const f32x4 w = *(const f32x4 *)(W + SX_W2 + n * SX_LD + k);
            a = fmaf(hf[k], w.x, a); a = fmaf(hf[k + 1], w.y, a); a = fmaf(hf[k + 2], w.z, a); a = fmaf(hf[k + 3], w.w, a);
        }
        h2[j] = a / (fabsf(a) + 1.0f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { hf[4 * j] = sx_quad<0>(h2[j]); hf[4 * j + 1] = sx_quad<1>(h2[j]); hf[4 * j + 2] = sx_quad<2>(h2[j]); hf[4 * j + 3] = sx_quad<3>(h2[j]); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = 4 * j + q;
        float a = W[SX_B3 + g];
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
            const f32x4 w = *(const f32x4 *)(W + SX_W3 + g * SX_LD + k);
            a = fmaf(hf[k], w.x, a); a = fmaf(hf[k + 1], w.y, a); a = fmaf(hf[k + 2], w.z, a); a = fmaf(hf[k + 3], w.w, a);
        }
        st[r][c][g] = (col * pr_pow8(a)) * 1.0000000000000001e-17f;
    }
    ro_fence();                     // the sixteen lanes of a row are in one wave: LDS writes above are visible to the reads below
    if (row >= M) continue;
    // lane (c, q) finishes g-point g = 4 c + q: absorption / Rayleigh depths of both humidity variants from the row's four combinations
    const int L = rowc / B, b = rowc - L * B, g = 4 * c + q;
    const size_t mrow = (size_t)L * (mem_B ? mem_B : B) + mem_off + b;        // row of the caller's (60, B_call, .) coin field
    float *o = S2 + (size_t)row * 48;
    float t_abs, t_sca;
    if (ngk > 0) {
        // sub-generation with k-point reductions (num11916, num87824): the coin picks the humidity variant per K-POINT (mask_u is
        // (60, B, ngk)), then tau_g = softplus(W tau_k + b) * 0.01 (+ 1e-9 for the absorption)
        const float *R1 = sred, *R2 = sred + SWX_RED_FLOATS;
        float a = R1[256 + g], sc = R2[256 + g];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool first = k < ngk ? mask_u[mrow * ngk + k] < 0.5f : true;
            const float ak = k < ngk ? (first ? st[r][0][k] : st[r][2][k]) : 0.0f, sk = k < ngk ? (first ? st[r][1][k] : st[r][3][k]) : 0.0f;
            a = fmaf(ak, R1[g * 16 + k], a); sc = fmaf(sk, R2[g * 16 + k], sc);
        }
        t_abs = pr_softplus(a) * 0.01f + 1.0000000000000001e-09f;
        t_sca = pr_softplus(sc) * 0.01f;
    } else {
        if (g >= ng) { o[g] = 1.0f; o[16 + g] = 0.0f; o[32 + g] = 0.0f; continue; }
        const bool first = mask_u[mrow * ng + g] < 0.5f;
        t_abs = fmaxf(first ? st[r][0][g] : st[r][2][g], 1.0000000000000001e-09f);
        t_sca = first ? st[r][1][g] : st[r][3][g];
    }
    float c_tau = 0.0f, c_sca = 0.0f, c_asy = 0.0f;
    if (L >= ilev) {
        const float *cs = CS + ((size_t)(L - ilev) * B + b) * 48;
        c_tau = cs[g]; c_sca = cs[16 + g]; c_asy = cs[32 + g];
    }
    const float tot = (t_abs + t_sca) + c_tau, sca = fmaxf(t_sca + c_sca, 1.0000000000000001e-09f);
    o[g] = tot; o[16 + g] = sca / tot; o[32 + g] = (c_asy * c_sca) / sca;
    }   // pass
}

#define RS_T 512
__global__ __launch_bounds__(RS_T) void phys_rad_solve_kernel(PhysDev d, int B, const float *__restrict__ x_sfc, const float *__restrict__ TP,
                                                            const float *__restrict__ CL, const float *__restrict__ S2,
                                                            const float *__restrict__ RS, float *__restrict__ out_lev, float *__restrict__ out_sfc)
{
    constexpr int L = PH_L, NG = PH_NG, NC = L * NG, NI = (L + 1) * NG;
    __shared__ float s_tr[NC], s_su[NC], s_sd[NC], s_pf[NC];                     // LW: transmittance, up / down source, Planck fraction
    __shared__ float s_R[NC], s_T[NC], s_Rd[NC], s_Td[NC], s_Tdir[NC];           // SW layer properties
    __shared__ float s_A[NI], s_Ad[NI];                                          // SW: albedo of everything below an interface
    __shared__ float s_ldn[NI], s_lup[NI], s_sup[NI], s_sdf[NI], s_sdr[NI];      // per g-point fluxes at the interfaces
    __shared__ float s_tl[L], s_pl[L], s_cd[L], s_ph[L + 1], s_bl[L + 1], s_net[L + 1], s_aux[32];
    const int b = blockIdx.x, tid = threadIdx.x, ilev = d.ilev;
    if (tid < d.naux) s_aux[tid] = x_sfc[(size_t)b * d.naux + tid] * d.xdiv_sca[tid] + d.xmean_sca[tid];
    __syncthreads();
    const float sp = s_aux[0];
    if (tid < L) {
        const size_t row = (size_t)tid * B + b;
        s_cd[tid] = RS[row * 2]; s_tl[tid] = RS[row * 2 + 1];
        s_pl[tid] = d.hyam[tid] * 100000.0f + sp * d.hybm[tid];
    }
    if (tid <= L) s_ph[tid] = sp * d.hybi[tid] + d.hyai[tid] * 100000.0f;
    __syncthreads();
    if (tid <= L) {                                   // interface temperatures (physics_rad.py:34) and their black-body flux
        const int j = tid;
        float t;
        if (j == 0) t = s_tl[0] + (s_ph[0] - s_pl[0]) * (s_tl[1] - s_tl[0]) / (s_pl[1] - s_pl[0]);
        else if (j == L) t = s_tl[L - 1] + (s_ph[L] - s_pl[L - 1]) * (s_tl[L - 1] - s_tl[L - 2]) / (s_pl[L - 1] - s_pl[L - 2]);
        else t = (s_pl[j - 1] * s_tl[j - 1] * (s_ph[j] - s_pl[j]) + s_pl[j] * s_tl[j] * (s_pl[j - 1] - s_ph[j])) / (s_ph[j] * (s_pl[j - 1] - s_pl[j]));
        const float t2 = t * t;
        s_bl[j] = t2 * t2 * 5.670374419e-8f;
    }
    const float mu0 = fmaxf(s_aux[6], 1e-6f);
    // ---- per (level, g-point) cell: LW optical depth + Planck fraction, SW two-stream coefficients ----
#pragma unroll                                       // (four trips: the global loads of all of them are issued before the first is used)
    for (int e0 = 0; e0 < NC; e0 += RS_T) {
        const int e = e0 + tid, lv = min(e >> 4, L - 1), g = e & 15;
        const bool ok = e < NC;
        const size_t row = (size_t)lv * B + b;
        const float *tp = TP + row * 32;
        const float logit = tp[16 + g], m = pr_max<NG>(logit), ex = expf(logit - m), pf = ex / pr_sum<NG>(ex);
        const float tau = pr_softplus(tp[g]) * 0.01f + (lv >= ilev ? CL[((size_t)(lv - ilev) * B + b) * NG + g] : 0.0f);
        const float od_lw = tau * 1.66f;
        const float *o = S2 + row * 48;
        // SW head: logits; SW gas-optics generation (d.swg): the final values
        const float od = d.swg ? o[g] : fminf(fmaxf(pr_pow8(o[g]) * (s_cd[lv] * 1e-23f), 1e-6f), 40.0f);
        const float ssa = d.swg ? o[16 + g] : pr_sigmoid(o[16 + g]), asy = d.swg ? o[32 + g] : pr_sigmoid(o[32 + g]);
        // two-stream coefficients (physics_rad.py:139)
        const float t_dir = expf(-od / mu0);
        const float g1 = (8.0f - ssa * (5.0f + 3.0f * asy)) * 0.25f, g2 = 3.0f * (ssa * (1.0f - asy)) * 0.25f;
        const float g3 = (2.0f - 3.0f * mu0 * asy) * 0.25f, g4 = 1.0f - g3;
        const float a1 = g1 * g4 + g2 * g3, a2 = g1 * g3 + g2 * g4;
        const float k = sqrtf(fmaxf((g1 - g2) * (g1 + g2), 1e-4f));
        const float ek = expf(-k * od), e2 = ek * ek, k2e = 2.0f * k * ek;
        float rf = 1.0f / (k + g1 + (k - g1) * e2);
        const float r_dif = g2 * (1.0f - e2) * rf;
        const float t_dif = fmaxf(fminf(fmaxf(k2e * rf, 0.0f), 1.0f - r_dif), 0.0f);
        const float kmu = k * mu0;
        float den = 1.0f - kmu * kmu;
        den = fabsf(den) > 1e-7f ? den : 1e-7f;
        rf = ssa * rf / den;
        const float kg3 = k * g3, kg4 = k * g4;
        float r_dir = rf * (((1.0f - kmu) * (a2 + kg3) - (1.0f + kmu) * (a2 - kg3) * e2) - k2e * (g3 - a2 * mu0) * t_dir);
        float t_dd = rf * (k2e * (g4 + a1 * mu0) - t_dir * ((1.0f + kmu) * (a1 + kg4) - (1.0f - kmu) * (a1 - kg4) * e2));
        const float room = 1.0f - t_dir;
        r_dir = fminf(fmaxf(r_dir, 0.0f), room);
        t_dd = fminf(fmaxf(t_dd, 0.0f), room - r_dir);
        if (ok) {
            s_pf[e] = pf; s_tr[e] = expf(-od_lw); s_su[e] = od_lw;
            s_R[e] = r_dif; s_T[e] = t_dif; s_Rd[e] = r_dir; s_Td[e] = t_dd; s_Tdir[e] = t_dir;
        }
    }
    __syncthreads();
    // LW layer sources (physics_rad.py:60): Planck flux of a g-point at the layer's top and bottom interfaces
    for (int e = tid; e < NC; e += RS_T) {
        const int lv = e >> 4;
        const float top = s_pf[e] * s_bl[lv];
        const float bot = lv < L - 1 ? s_pf[e + NG] * s_bl[lv + 1] : s_pf[e] * s_bl[L];
        const float c = s_su[e] * 0.2f, up = (1.0f - s_tr[e]) * ((top + bot) * 0.5f + c * top) / (c + 1.0f);
        s_su[e] = up;
        // the first exports hand the UPWARD source to the downward sweep as well (an aliased view in the serialised graph)
        s_sd[e] = d.lw_dn ? (1.0f - s_tr[e]) * ((top + bot) * 0.5f + c * bot) / (c + 1.0f) : up;
    }
    __syncthreads();
    // ---- level recurrences, one lane per g-point: LW on wave 0, SW on wave 1 ----
    if (tid < NG) {
        const int g = tid;
        float f = 0.0f;
        s_ldn[g] = 0.0f;
#pragma unroll 4                                      // (the LDS operands of the next levels are fetched ahead of the dependent chain)
        for (int j = 0; j < L; ++j) {
            f = s_tr[j * NG + g] * f + s_sd[j * NG + g];
            s_ldn[(j + 1) * NG + g] = f;
        }
        f = s_pf[(L - 1) * NG + g] * s_aux[11];       // surface emission (emissivity 1)
        s_lup[L * NG + g] = f;
#pragma unroll 4
        for (int j = L - 1; j >= 0; --j) {
            f = s_tr[j * NG + g] * f + s_su[j * NG + g];
            s_lup[j * NG + g] = f;
        }
    } else if (tid >= 64 && tid < 64 + NG) {
        const int g = tid - 64;
        const int n_ir = d.n_ir, n_mix = d.n_mix;     // near-IR | mixed | visible g-points (11, 13 of 16 in the unfrozen graphs)
        const float toa = s_aux[1] * d.toa_spec[g];
        float A, Ad;
        if (d.nx21) {       // learned (or 0.5 / 0.5) weights of the mixed g-points, in the export's operation order
            A = g < n_ir ? s_aux[7] : g < n_mix ? d.mix_near * s_aux[7] + d.mix_vis * s_aux[9] : s_aux[9];
            Ad = g < n_ir ? s_aux[8] : g < n_mix ? d.mix_near * s_aux[8] + d.mix_vis * s_aux[10] : s_aux[10];
        } else {
            A = g < n_ir ? s_aux[7] : g < n_mix ? (s_aux[7] + s_aux[9]) * 0.5f : s_aux[9];
            Ad = g < n_ir ? s_aux[8] : g < n_mix ? (s_aux[8] + s_aux[10]) * 0.5f : s_aux[10];
        }
        const float lo = d.nx21 ? -3.0e38f : 0.0f;    // the unfrozen graphs clip the SW fluxes at zero, the exports do not
        s_A[L * NG + g] = A; s_Ad[L * NG + g] = Ad;
#pragma unroll 4
        for (int j = L - 1; j >= 0; --j) {
            const int e = j * NG + g;
            // (the division is the dependent chain of this sweep: v_rcp_f32, 1 ulp, instead of the ten-instruction IEEE sequence)
            const float Tj = s_T[e], inv = __builtin_amdgcn_rcpf(1.0f - A * s_R[e]);
            Ad = s_Rd[e] + (s_Tdir[e] * Ad + s_Td[e] * A) * Tj * inv;
            A = s_R[e] + (Tj * Tj) * A * inv;
            s_A[e] = A; s_Ad[e] = Ad;
        }
        float dif = 0.0f, dr = toa;
        s_sup[g] = fmaxf(toa * Ad, lo); s_sdf[g] = 0.0f; s_sdr[g] = fmaxf(toa, lo);
#pragma unroll 4
        for (int j = 0; j < L; ++j) {
            const int e = j * NG + g;
            const float Ab = s_A[e + NG], Adb = s_Ad[e + NG];
            const float inv = __builtin_amdgcn_rcpf(1.0f - s_R[e] * Ab);      // (off the chain: known once the upward sweep is done)
            dif = (s_T[e] * dif + dr * (s_T[e] * Adb * s_R[e] + s_Td[e])) * inv;
            dr = dr * s_Tdir[e];
            s_sup[e + NG] = fmaxf(dr * Adb + dif * Ab, lo);
            s_sdf[e + NG] = fmaxf(dif, lo);
            s_sdr[e + NG] = fmaxf(dr, lo);
        }
    }
    __syncthreads();
    // ---- spectral sums, net flux per interface, surface diagnostics ----
    const bool day = !(s_aux[6] < 1e-6f);           // night columns: shortwave terms are SET to zero (assignment upstream, so no 0 * inf)
    if (tid <= L) {
        const int j = tid;
        float ldn = 0.0f, lup = 0.0f, up = 0.0f, df = 0.0f, dr = 0.0f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            ldn += s_ldn[j * NG + g]; lup += s_lup[j * NG + g];
            up += s_sup[j * NG + g]; df += s_sdf[j * NG + g]; dr += s_sdr[j * NG + g];
        }
        const float sw_dn = df + dr;
        s_net[j] = (ldn - lup) + (day ? sw_dn - up : 0.0f);
        if (j == L) {
            float dir_ir = 0.0f, dir_mix = 0.0f, dir_vis = 0.0f, dif_ir = 0.0f, dif_mix = 0.0f, dif_vis = 0.0f;
            for (int g = 0; g < NG; ++g) {
                const float a = s_sdr[L * NG + g], c = s_sdf[L * NG + g];
                if (g < d.n_ir) { dir_ir += a; dif_ir += c; } else if (g < d.n_mix) { dir_mix += a; dif_mix += c; } else { dir_vis += a; dif_vis += c; }
            }
            float *os = out_sfc + (size_t)b * 8;
            os[0] = (day ? (d.nx21 && !d.sfc_sw_down ? sw_dn - up : sw_dn) : 0.0f) * d.ys_rad[0];     // (most exports return the NET surface shortwave)
            os[1] = ldn * d.ys_rad[1];
            os[4] = (day ? dir_vis + d.mix_vis * dir_mix : 0.0f) * d.ys_rad[2];      // SOLS
            os[5] = (day ? dir_ir + d.mix_near * dir_mix : 0.0f) * d.ys_rad[3];      // SOLL
            os[6] = (day ? dif_vis + d.mix_vis * dif_mix : 0.0f) * d.ys_rad[4];      // SOLSD
            os[7] = (day ? dif_ir + d.mix_near * dif_mix : 0.0f) * d.ys_rad[5];      // SOLLD
        }
    }
    __syncthreads();
    if (tid < L) {
        const int j = tid;
        const float pd = sp * (d.hybi[j + 1] - d.hybi[j]) + (d.hyai[j + 1] - d.hyai[j]) * 100000.0f;
        const float dT = -((s_net[j + 1] - s_net[j]) / pd) * 0.009761357302f * d.yscale_lev[j * 5];
        out_lev[((size_t)b * L + j) * 5] += dT;
    }
}

int launch_phys_radiation(csa_phys *h, int B, const float *x_sfc, float *out_lev, float *out_sfc, hipStream_t s, const float *mask_u)
{
    const PhysDev &d = h->d;
    const int M = PH_L * B;
    RadOptics a{h->XG, h->XR, h->RS, h->g_w1, h->g_b1, h->g_w2, h->g_b2, h->g_w3, h->g_b3, h->r1_w, h->r1_b, h->r2_w, h->r2_b,
                h->s1_w, h->s1_b, h->s2_w, h->s2_b, d.g_ystd, d.g_ymean, h->TP, h->S2, M};
    // grid y: 0 = LW gas optics, 1 = SW head (absent in the SW gas-optics generation, which has its own kernel)
    hipLaunchKernelGGL(rad_optics_kernel, dim3((M + 32 * RO_WAVES - 1) / (32 * RO_WAVES), d.swg ? 1 : 2), dim3(64 * RO_WAVES), 0, s, a);
    CSA_HIP_CHECK(hipGetLastError());
    if (d.nx21 && d.swg && !d.sw_e3sm) {
        if (!mask_u) { csa_set_error_msg("physRNN (frozen export): the SW humidity coin needs its uniform draws"); return CSA_ERR_ARG; }
        const int npass = M >= 60 * 1024 ? 4 : 1;
        hipLaunchKernelGGL(rad_sw_gas16_kernel, dim3((M + SX_ROWS * npass - 1) / (SX_ROWS * npass)), dim3(256), 0, s, h->XR, d.swg, h->CS, mask_u, h->S2, M, B,
                           d.ilev, h->ng, d.sw_ngk, npass, d.mem_B, d.mem_off);
        CSA_HIP_CHECK(hipGetLastError());
    } else if (d.swg) {
        hipLaunchKernelGGL(rad_sw_gas_kernel, dim3((M + 32 * SG_WAVES - 1) / (32 * SG_WAVES)), dim3(64 * SG_WAVES), 0, s, h->XR, d.swg, h->CS, h->S2, M, B, d.ilev);
        CSA_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(phys_rad_solve_kernel, dim3(B), dim3(RS_T), 0, s, d, B, x_sfc, h->TP, h->CL, h->S2, h->RS, out_lev, out_sfc);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
