// gemm.hip -- hoisted input projection  C(M,N) = A(M,K) * W(N,K)^T + bias(N), exact fp32 on
// the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, no reduced precision).
//
// This is the non-recurrent half of nn.LSTM / nn.GRU (W_ih x_t + b for all 60 levels at once,
// rnn/models/models.py:493,536), the same hoisting the reference applies to its own fused GRU
// (rnn/models_torch_kernels.py:858-862).  M = nlev*B rows (sequence-major), N = G*nh, K = nh(+nh_mem).
//
// Tiling: 128x128 block tile, 4 waves (2x2), each wave a 64x64 tile = 2x2 MFMA 32x32 tiles
// (64 accumulator VGPRs), K consumed in chunks of 16 staged through LDS.  A and W are both
// "row-major with k contiguous", so one staging routine serves both operands; rows are padded
// to KC+1 floats so that the MFMA operand reads (lane = row, fixed k) are bank-conflict free.
// Global loads for chunk c+1 are issued before the MFMAs of chunk c and land in the other LDS
// buffer after them (register + LDS double buffer, one barrier per chunk).
#include "common.h"

#ifndef CSA_SMALL_GEMM_ROWS_DEFAULT
#define CSA_SMALL_GEMM_ROWS_DEFAULT 11520   // 192 columns x 60 levels: measured crossover (tools/latency_sweep.py)
#endif
#define GB_M 128
#define GB_N 128
#define GB_K 16
#define GB_LD (GB_K + 1)
#define GB_THREADS 256

// MI = 32-row MFMA sub-tiles per wave in M: 2 -> 128x128 block tile (default), 1 -> 64x128 (problems whose 128-row
// tiling would leave CUs without work: the input-gradient GEMMs of training, N <= 144, K = 512)
template <int MI>
__global__ __launch_bounds__(GB_THREADS) void proj_gemm_kernel(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    float *__restrict__ C, int M, int N, int K, int tiles_m, int tiles_n, int act, float alpha, int n_lin,
    int lda, int ldc, int conv_L, int conv_cin, GemmEpi epi)
{
    // epi (training of the CNN baseline, cnn_train.hip): after bias + activation,
    //   mask   : v *= mask[row*ldmask + col] ? mscale : 0      (inverted dropout, hpo_train.py:169,177)
    //   gate   : v *= gate[row*ldc + col] > 0 ? gscale : 0     (backward of ReLU + dropout from the SAVED output)
    //   addsrc : v += addsrc[row*ldc + col]                    (residual add; addsrc == C accumulates in place)
    // lda / ldc: leading dimensions of A and C (K and N for a plain GEMM).
    // conv_L > 0: implicit-GEMM 1-D convolution, kernel 3, 'same' padding, over rows = (column, level) with
    //   conv_L levels per column: A row r is the 3*cin contiguous floats [x(l-1), x(l), x(l+1)] starting at
    //   A + (r-1)*lda (lda = cin), with the first / last third masked to zero on the first / last level
    //   (baseline_models/CNN/training/hpo_train.py:165-177, Conv1D(..., padding="same")).
    // accumulate: C += result (the 1x1 residual projection added onto the block output, :180-184).
    // act 3: ELU (the pre-output Conv1D(10, 1, activation="elu"), :189-194).
    // act: 0 none; 1 LeakyReLU(alpha) on every column; 2 split head: columns < n_lin linear, the rest ReLU
    // (the Keras MLP baseline's Dense(120,linear) || Dense(8,relu) output, step2_retrain.py:118-121)
    // double-buffered LDS: chunk c+1 is written while chunk c is being multiplied -> one barrier
    // per 16-deep K chunk (8 k-steps x 4 MFMA = 2048 MFMA cycles per wave between barriers)
    constexpr int TM = 64 * MI;
    __shared__ float As[2][TM * GB_LD];
    __shared__ float Ws[2][GB_N * GB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of tiles and walk N fastest inside it: the tiles_n blocks that share an A
    // row-panel then sit on one XCD and re-read it from that XCD's L2 (speed only).
    int bid = blockIdx.x;
    {
        const int nwg = tiles_m * tiles_n;
        if (nwg % 8 == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    }
    const int m0 = (bid / tiles_n) * TM, n0 = (bid % tiles_n) * GB_N;

    // staging map: thread -> (row r, r+64; k quad kq)
    const int lr = tid >> 2, kq = (tid & 3) * 4;
    f32x4 ra[2], rw[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + lr + 64 * i, col = n0 + lr + 64 * i, k = k0 + kq;
            bool ok = row < M && k < K;
            if (conv_L > 0 && ok) {
                const int l = row % conv_L;
                ok = !((l == 0 && k < conv_cin) || (l == conv_L - 1 && k >= 2 * conv_cin));
            }
            const float *ap = conv_L > 0 ? A + ((long)row - 1) * lda + k : A + (size_t)row * lda + k;
            if (i < MI) ra[i] = ok ? *(const f32x4 *)ap : f32x4{0, 0, 0, 0};
            rw[i] = (col < N && k < K) ? *(const f32x4 *)(W + (size_t)col * K + k) : f32x4{0, 0, 0, 0};
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float *pa = As[buf] + (lr + 64 * i) * GB_LD + kq;
            float *pw = Ws[buf] + (lr + 64 * i) * GB_LD + kq;
            if (i < MI) { pa[0] = ra[i].x; pa[1] = ra[i].y; pa[2] = ra[i].z; pa[3] = ra[i].w; }
            pw[0] = rw[i].x; pw[1] = rw[i].y; pw[2] = rw[i].z; pw[3] = rw[i].w;
        }
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int arow = (wm * 32 * MI + (lane & 31)) * GB_LD + (lane >> 5);
    const int wrow = (wn * 64 + (lane & 31)) * GB_LD + (lane >> 5);

#ifdef GEMM_EXP_STAGGER   /* diagnostic: de-phase co-resident workgroups (convoy test) */
    {
        const int ph = (blockIdx.x / GEMM_EXP_STAGGER) % 3;
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)ph * 6000ull) __builtin_amdgcn_s_sleep(8);
    }
#endif
    gload(0);
    sstore(0);
    __syncthreads();
    const int nchunk = (K + GB_K - 1) / GB_K;
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
#ifndef GEMM_EXP_NO_GLOAD
        if (c + 1 < nchunk) gload((c + 1) * GB_K);
#endif
        const float *as = As[cur], *ws = Ws[cur];
#pragma unroll
        for (int kk = 0; kk < GB_K / 2; ++kk) {
#ifdef GEMM_EXP_NO_LDSREAD   /* diagnostic builds only (tools/): wrong results, timing of the remaining parts */
            const float a0 = 1e-3f * (lane + kk + c), a1 = a0 + 1.0f, b0 = a0 * 0.5f, b1 = a0 + 2.0f;
#else
            const float a0 = as[arow + kk * 2], a1 = MI > 1 ? as[arow + 32 * GB_LD + kk * 2] : 0.0f;
            const float b0 = ws[wrow + kk * 2], b1 = ws[wrow + 32 * GB_LD + kk * 2];
#endif
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            if constexpr (MI > 1) {
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
#ifndef GEMM_EXP_NO_GLOAD
        if (c + 1 < nchunk) sstore(cur ^ 1);
#endif
#ifndef GEMM_EXP_NO_BARRIER
        __syncthreads();
#endif
    }

    // epilogue: D[i][j] with j = lane&31 (column) and i = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // The bias is added to every accumulator BEFORE the first store: a load consumed between
    // stores makes hipcc emit s_waitcnt vmcnt(0) there, which also waits for the stores already
    // issued and serialises the whole 64-store epilogue (measured: +9 us per launch).
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        bv[j] = (bias && col < N) ? bias[col] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bv[j];
                if (act == 1) v = v > 0.0f ? v : alpha * v;
                else if (act == 2 && n0 + wn * 64 + j * 32 + (lane & 31) >= n_lin) v = fmaxf(v, 0.0f);
                else if (act == 3) v = v > 0.0f ? v : expm1f(v);
                acc[i][j][r] = v;
            }
#ifdef GEMM_EXP_NO_STORE   /* diagnostic: keep the accumulators alive, store one value per wave */
    {
        float t = 0.0f;
        for (int j = 0; j < 2; ++j) for (int i = 0; i < MI; ++i) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 123.456f) C[0] = t;
        return;
    }
#endif
    const bool plain = !epi.addsrc && !epi.mask && !epi.gate;
    const bool interior = (m0 + TM <= M) && (n0 + GB_N <= N) && plain;
    if (interior) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float *cp = C + (size_t)(m0 + wm * 32 * MI + 4 * (lane >> 5)) * ldc + n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    cp[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc] = acc[i][j][r];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < M && col < N) {
                        const size_t idx = (size_t)row * ldc + col;
                        float v = acc[i][j][r];
                        if (epi.mask) v = (col < epi.ldmask && epi.mask[(size_t)row * epi.ldmask + col]) ? v * epi.mscale : 0.0f;
                        if (epi.gate) v *= epi.gate[idx] > 0.0f ? epi.gscale : epi.gneg;
                        if (epi.addsrc) v += epi.addsrc[idx];
                        C[idx] = v;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Split-operand variant of the plain projection (opt-in: csa_set_gemm_split / CSA_GEMM_SPLIT_BF16=1; the default stays
// the fp32 MFMA chain above).  Every fp32 operand is written as hi + mid + lo, three bf16 values (v_cvt_pk_bf16_f32,
// round-to-nearest: the split is EXACT, 3 x 8 significand bits), and the product a*b is accumulated in fp32 from the six
// partial products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid on v_mfma_f32_32x32x16_bf16 (bf16 x bf16 is exact in
// fp32).  Dropped: mid*lo, lo*mid, lo*lo <= 2^-24 |a||b| each -- the size of ONE fp32 rounding, of which the fp32 chain
// makes K.  Six bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles per 16-deep K step (2.7x fewer matrix-pipe
// cycles).  Same 128x128 tile / 2x2 waves / LDS double buffer; operands are split ONCE, while staging, into three bf16
// planes per operand.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define B3_LD 16
// LDS rows are 16 bf16 = 32 bytes, unpadded (49 KB per workgroup: THREE workgroups per CU, so the 720 tiles of the headline shape
// run in one round); the two 16-byte halves of rows 8..15 (mod 16) are swapped, which keeps the 16-byte operand reads of 16
// consecutive rows on distinct banks
__device__ __forceinline__ int b3_swz(int row) { return (row >> 3) & 1; }
__device__ __forceinline__ void b3_split(const f32x4 v, bf16x4 &h, bf16x4 &m, bf16x4 &l)
{
    h = __builtin_convertvector(v, bf16x4);
    const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
    m = __builtin_convertvector(r1, bf16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);
    l = __builtin_convertvector(r2, bf16x4);
}
__global__ __launch_bounds__(GB_THREADS, 3) void proj_gemm_b3_kernel(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    float *__restrict__ C, int M, int N, int K, int tiles_m, int tiles_n)
{
    __shared__ __attribute__((aligned(16))) __bf16 As[2][3][GB_M * B3_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[2][3][GB_N * B3_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = blockIdx.x;
    {
        const int nwg = tiles_m * tiles_n;
        if (nwg % 8 == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    }
    const int m0 = (bid / tiles_n) * GB_M, n0 = (bid % tiles_n) * GB_N;
    const int lr = tid >> 2, kq = (tid & 3) * 4;
    // three-stage pipeline: the global loads of chunk c+2 are in flight (asm-issued, counted s_waitcnt) while chunk c is multiplied
    // and chunk c+1 (loaded one iteration earlier) is split and written to the other LDS buffer between the MFMAs of chunk c
    f32x4 ra[2][2], rw[2][2];
    const float *pa[2], *pw[2];
    bool rok[2], cok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + lr + 64 * i, col = n0 + lr + 64 * i;
        rok[i] = row < M; cok[i] = col < N;
        pa[i] = A + (size_t)min(row, M - 1) * K;
        pw[i] = W + (size_t)min(col, N - 1) * K;
    }
#define B3_GLOAD(k0, SET)                                                                                             \
    {                                                                                                                 \
        const int kc = min((k0) + kq, K - 4);          /* clamped address; values past K are zeroed when they are split */ \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ra[SET][0]) : "v"(pa[0] + kc));                        \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ra[SET][1]) : "v"(pa[1] + kc));                        \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(rw[SET][0]) : "v"(pw[0] + kc));                        \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(rw[SET][1]) : "v"(pw[1] + kc));                        \
    }
    // split one float4 of chunk k0 (register R, valid flag OK) into the three planes of operand buffer P at offset o
#define B3_PUT(P, R, OK, k0, o)                                                                                       \
    {                                                                                                                 \
        f32x4 v_ = R;                                                                                                 \
        if (!((OK) && (k0) + kq < K)) v_ = f32x4{0, 0, 0, 0};                                                         \
        bf16x4 h_, m_, l_;                                                                                            \
        b3_split(v_, h_, m_, l_);                                                                                     \
        *(bf16x4 *)(P[0] + (o)) = h_; *(bf16x4 *)(P[1] + (o)) = m_; *(bf16x4 *)(P[2] + (o)) = l_;                     \
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // operand element e of lane (row = lane & 31, group g = lane >> 5) is k = 8 g + e, for A and W alike
    const int arow = (wm * 64 + (lane & 31)) * B3_LD + 8 * ((lane >> 5) ^ b3_swz(lane & 31));
    const int wrow = (wn * 64 + (lane & 31)) * B3_LD + 8 * ((lane >> 5) ^ b3_swz(lane & 31));
    const int o0 = lr * B3_LD + 8 * ((kq >> 3) ^ b3_swz(lr)) + (kq & 7), o1 = o0 + 64 * B3_LD;
    const int nchunk = (K + GB_K - 1) / GB_K;
    B3_GLOAD(0, 0)
    B3_GLOAD(GB_K, 1)
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(rw[0][0]), "+v"(rw[0][1]));
    B3_PUT(As[0], ra[0][0], rok[0], 0, o0) B3_PUT(As[0], ra[0][1], rok[1], 0, o1)
    B3_PUT(Ws[0], rw[0][0], cok[0], 0, o0) B3_PUT(Ws[0], rw[0][1], cok[1], 0, o1)
    __syncthreads();
#define B3_MMA(i, j)                                                                                                  \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);                        \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);                        \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);                        \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);                        \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);                        \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
    // chunk c from LDS buffer c & 1; SET holds chunk c+1; chunk c+2 goes into the other set (consumed one iteration ago).
    // Six partial products per tile, smallest first; the split of one float4 follows each tile's MFMAs in program order.
#define B3_CHUNK(c, SET)                                                                                              \
    {                                                                                                                 \
        const int cur = (c) & 1, k1 = ((c) + 1) * GB_K;                                                               \
        B3_GLOAD(((c) + 2) * GB_K, (SET) ^ 1)                                                                         \
        bf16x8 a[2][3], b[2][3];                                                                                      \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                                 \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                           \
                a[i][p] = *(const bf16x8 *)(As[cur][p] + arow + i * 32 * B3_LD);                                      \
                b[i][p] = *(const bf16x8 *)(Ws[cur][p] + wrow + i * 32 * B3_LD);                                      \
            }                                                                                                         \
        B3_MMA(0, 0)                                                                                                  \
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(ra[SET][0]), "+v"(ra[SET][1]), "+v"(rw[SET][0]), "+v"(rw[SET][1])); \
        B3_PUT(As[cur ^ 1], ra[SET][0], rok[0], k1, o0)                                                               \
        B3_MMA(0, 1)                                                                                                  \
        B3_PUT(As[cur ^ 1], ra[SET][1], rok[1], k1, o1)                                                               \
        B3_MMA(1, 0)                                                                                                  \
        B3_PUT(Ws[cur ^ 1], rw[SET][0], cok[0], k1, o0)                                                               \
        B3_MMA(1, 1)                                                                                                  \
        B3_PUT(Ws[cur ^ 1], rw[SET][1], cok[1], k1, o1)                                                               \
        __syncthreads();                                                                                              \
    }
    for (int c = 0; c < nchunk; c += 2) {      // (an odd chunk count runs one more chunk of zeros)
        B3_CHUNK(c, 1)
        B3_CHUNK(c + 1, 0)
    }
    // the last two (unused) prefetches: their target registers stay allocated until the loads have landed -- tied to the wait, or
    // the epilogue could be given a register that an in-flight load still writes
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(rw[0][0]), "+v"(rw[0][1]),
                                        "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(rw[1][0]), "+v"(rw[1][1]) :: "memory");
#undef B3_CHUNK
#undef B3_MMA
#undef B3_PUT
#undef B3_GLOAD
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        bv[j] = (bias && col < N) ? bias[col] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += bv[j];
    if ((m0 + GB_M <= M) && (n0 + GB_N <= N)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float *cp = C + (size_t)(m0 + wm * 64 + 4 * (lane >> 5)) * N + n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    cp[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * N] = acc[i][j][r];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
                }
        }
    }
}

static int g_gemm_split = -1;      // 1: the plain projection runs on the split-bf16 kernel (csa_set_gemm_split)
extern "C" int csa_set_gemm_split(int on)
{
    g_gemm_split = on != 0;
    return CSA_OK;
}

static int g_small_rows = -1;      // rows up to which the projection uses the small-M kernel (csa_set_small_gemm_rows)
extern "C" int csa_set_small_gemm_rows(int rows)
{
    if (rows < 0) return CSA_ERR_ARG;
    g_small_rows = rows;
    return CSA_OK;
}

int launch_proj_gemm(const float *A, const float *W, const float *bias, float *C,
                     int M, int N, int K, hipStream_t s, int class_rows)
{
    // class_rows: the row count the kernel class is chosen for (a column half inherits the class of the whole call)
    if (g_small_rows < 0) {
        const char *e = getenv("CSA_SMALL_GEMM_ROWS");
        g_small_rows = e ? atoi(e) : CSA_SMALL_GEMM_ROWS_DEFAULT;
    }
    if ((class_rows > 0 ? class_rows : M) <= g_small_rows) return launch_gemm_small(A, W, bias, C, M, N, K, 0, 0.0f, 0, s);
    if (g_gemm_split < 0) g_gemm_split = getenv("CSA_GEMM_SPLIT_BF16") ? atoi(getenv("CSA_GEMM_SPLIT_BF16")) != 0 : 0;
    // (wide outputs only: the forward projections, N = 3 nh / 4 nh -- what was measured and parity-tested; the narrow input-gradient
    // GEMMs of training, N = nh, keep the fp32 kernel and its 64-row tiles)
    if (g_gemm_split && K % 4 == 0 && N >= 256) {
        const int tiles_m = (M + GB_M - 1) / GB_M, tiles_n = (N + GB_N - 1) / GB_N;
        hipLaunchKernelGGL(proj_gemm_b3_kernel, dim3(tiles_m * tiles_n), dim3(GB_THREADS), 0, s, A, W, bias, C, M, N, K, tiles_m, tiles_n);
        CSA_HIP_CHECK(hipGetLastError());
        return CSA_OK;
    }
    return launch_gemm_act(A, W, bias, C, M, N, K, 0, 0.0f, 0, s);
}

int launch_gemm_act(const float *A, const float *W, const float *bias, float *C, int M, int N, int K,
                    int act, float alpha, int n_lin, hipStream_t s)
{
    return launch_gemm_ex(A, W, bias, C, M, N, K, act, alpha, n_lin, K, N, 0, 0, 0, s);
}

int launch_gemm_ex(const float *A, const float *W, const float *bias, float *C, int M, int N, int K, int act, float alpha,
                   int n_lin, int lda, int ldc, int conv_L, int conv_cin, int accumulate, hipStream_t s)
{
    GemmEpi e{};
    if (accumulate) e.addsrc = C;
    return launch_gemm_epi(A, W, bias, C, M, N, K, act, alpha, n_lin, lda, ldc, conv_L, conv_cin, e, s);
}

int launch_gemm_epi(const float *A, const float *W, const float *bias, float *C, int M, int N, int K, int act, float alpha,
                    int n_lin, int lda, int ldc, int conv_L, int conv_cin, const GemmEpi &epi, hipStream_t s)
{
    if ((lda % 4) || (conv_L > 0 && (conv_cin % 4 || K != 3 * conv_cin))) {
        csa_set_error_msg("gemm_ex: lda and cin must be multiples of 4, K = 3*cin in conv mode");
        return CSA_ERR_UNSUPPORTED;
    }
    if (K % 4 != 0) {
        csa_set_error_msg("proj_gemm: K must be a multiple of 4");
        return CSA_ERR_UNSUPPORTED;
    }
    int tiles_m = (M + GB_M - 1) / GB_M;
    const int tiles_n = (N + GB_N - 1) / GB_N;
    // narrow outputs (N <= 384) whose 128-row tiling gives the 256 CUs fewer than two or three workgroups each: 64-row tiles
    static const int m64 = getenv("CSA_GEMM_M64") ? atoi(getenv("CSA_GEMM_M64")) : 1;
    // (three N-tiles: the unpadded GRU projection, N = 3 nh = 384 -- 183.8 -> 179.7 us per cur_gru128 call with 64-row tiles)
    // (four N-tiles, the LSTM projection at 23,040 rows: 64-row tiles measured 0.2044 against 0.2037 ms per step -- not taken)
    const bool half_m = m64 && ((tiles_n <= 2 && tiles_m * tiles_n < 512) || (tiles_n == 3 && tiles_m * tiles_n < 768)) && M > 64;
    if (half_m) tiles_m = (M + 63) / 64;
#ifdef GEMM_EXP_EXTRA_LDS
    static const int extra_lds = getenv("CSA_GEMM_EXTRA_LDS") ? atoi(getenv("CSA_GEMM_EXTRA_LDS")) : 0;   // diagnostic: caps occupancy
#else
    const int extra_lds = 0;
#endif
    if (half_m)
        hipLaunchKernelGGL(proj_gemm_kernel<1>, dim3(tiles_m * tiles_n), dim3(GB_THREADS), extra_lds, s, A, W, bias, C, M, N, K,
                           tiles_m, tiles_n, act, alpha, n_lin, lda, ldc, conv_L, conv_cin, epi);
    else
        hipLaunchKernelGGL(proj_gemm_kernel<2>, dim3(tiles_m * tiles_n), dim3(GB_THREADS), extra_lds, s, A, W, bias, C, M, N, K,
                           tiles_m, tiles_n, act, alpha, n_lin, lda, ldc, conv_L, conv_cin, epi);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Small-M variant (the MLP baseline at a few hundred rows): the 128x128 tiling above leaves a 384-row problem with
// 3 x N/128 workgroups on a 256-CU part and a K-long serial MFMA chain in each.  Here a workgroup owns one 32x32
// output tile and its four waves split K between them (each wave one MFMA accumulator, operands straight from
// global memory as float4: lanes < 32 take k..k+3 and lanes >= 32 take k+4..k+7 of a row, i.e. MFMA e contracts
// the pair (k+e, k+4+e)); the four partial tiles are summed through LDS in a fixed order (deterministic), then bias +
// activation.  384 x 768: 288 workgroups, each wave K/8 MFMAs.
template <int PD>
__global__ __launch_bounds__(256) void gemm_small_kernel(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ C,
    int M, int N, int K, int act, float alpha, int n_lin)
{
    __shared__ float red[3][32 * 33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int r = lane & 31, half = lane >> 5;
    const bool am = m0 + r < M, wn = n0 + r < N;
    const float *ap = A + (size_t)(am ? m0 + r : 0) * K + 4 * half;
    const float *wp = W + (size_t)(wn ? n0 + r : 0) * K + 4 * half;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const int nch = (K + 7) / 8;                     // chunks of 8 k-values, dealt round-robin to the waves
    // rolling prefetch, PD chunks deep: the operands come straight from global memory (L2), so the loop is a chain of
    // load -> 4 MFMA.  PD = 4 for long K (MLP, K up to 768: 84 -> 67 us per 384-row forward), 1 for the K <= 160 projections
    // (a wave has only 4-5 chunks there and the extra registers cost more than the prefetch gains)
    f32x4 a[PD], w[PD];
    auto fetch = [&](int c, f32x4 &av, f32x4 &wv) {
        av = f32x4{0, 0, 0, 0};
        wv = f32x4{0, 0, 0, 0};
        if (c < nch && 8 * c + 4 * half < K) {
            if (am) av = *(const f32x4 *)(ap + 8 * c);
            if (wn) wv = *(const f32x4 *)(wp + 8 * c);
        }
    };
#pragma unroll
    for (int d = 0; d < PD; ++d) fetch(wave + 4 * d, a[d], w[d]);
    for (int c = wave; c < nch; c += 4 * PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const f32x4 ac = a[d], wc = w[d];
            fetch(c + 4 * (d + PD), a[d], w[d]);
            if (c + 4 * d < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[e], wc[e], acc, 0, 0, 0);
            }
        }
    }
    // acc[i]: row (i&3) + 8*(i>>2) + 4*half, column r of the tile
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave - 1][((i & 3) + 8 * (i >> 2) + 4 * half) * 33 + r] = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
        const int col = n0 + r;
        const float bv = (bias && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = (i & 3) + 8 * (i >> 2) + 4 * half, row = m0 + rr;
            float v = ((acc[i] + red[0][rr * 33 + r]) + (red[1][rr * 33 + r] + red[2][rr * 33 + r])) + bv;
            if (act == 1) v = v > 0.0f ? v : alpha * v;
            else if (act == 2 && col >= n_lin) v = fmaxf(v, 0.0f);
            else if (act == 3) v = v > 0.0f ? v : expm1f(v);
            if (row < M && col < N) C[(size_t)row * N + col] = v;
        }
    }
}

int launch_gemm_small(const float *A, const float *W, const float *bias, float *C, int M, int N, int K, int act, float alpha,
                      int n_lin, hipStream_t s)
{
    if (K % 4) { csa_set_error_msg("gemm_small: K must be a multiple of 4"); return CSA_ERR_UNSUPPORTED; }
    if (K >= 256)
        hipLaunchKernelGGL(gemm_small_kernel<4>, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, s, A, W, bias, C, M, N, K, act, alpha, n_lin);
    else
        hipLaunchKernelGGL(gemm_small_kernel<1>, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, s, A, W, bias, C, M, N, K, act, alpha, n_lin);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
