// Round-3 experiment, NOT part of the library (fifth negative result on the projection GEMM's store drain): persistent
// 128 x 128 tiles with the next tile's first K-chunk requested in the middle of the 64-store epilogue.  Built into gemm.hip behind
// CSA_GEMM_PERSIST = <workgroups> it measured, on v4_memory_2700 (two 1,350-column halves on two streams; baseline 0.985 ms):
//   occupancy 2 (109 VGPR + 124 AGPR, the compiler's choice): 512 / 768 / 1,024 workgroups 1.087 / 1.081 / 1.076 ms
//   occupancy 3 (amdgpu_waves_per_eu(3, 3), 122 VGPR):        384 / 768 / 1,024 / 1,536      1.026 / 1.108 / 1.016 / 0.998 ms
//   one stream (--halves 0; baseline 1.038 ms):                768 / 1,536                    1.050 / 1.056 ms
// i.e. the fewer tiles a workgroup walks, the closer it gets to the non-persistent kernel, and never below it: at three workgroups
// per CU the hardware already overlaps one workgroup's epilogue with its neighbours' matrix phases, and the dispatcher's refill is
// cheaper than the bookkeeping registers of the walk.
// ------------------------------------------------------------------------------------------------------------------
// Persistent variant of the plain projection (bias only, no activation / epilogue extras / conv mode), 128 x 128 tiles: a workgroup
// walks tiles blockIdx.x, + gridDim.x, ...; the first K-chunk of the NEXT tile is requested in the MIDDLE of this tile's 64 stores
// per thread, so that its latency and the second half of the store drain lie behind the next tile's matrix phase (the compiler's
// vmcnt for the prefetched registers then allows the 32 younger stores to stay in flight).  DESIGN 4.2; selected by
// CSA_GEMM_PERSIST = number of workgroups.
__global__ __launch_bounds__(GB_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void proj_gemm_persist_kernel(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ C,
    int M, int N, int K, int tiles_m, int tiles_n, int lda, int ldc)
{
    __shared__ float As[2][GB_M * GB_LD];
    __shared__ float Ws[2][GB_N * GB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int nwg = tiles_m * tiles_n, step = gridDim.x;
    const int lr = tid >> 2, kq = (tid & 3) * 4;
    const int arow = (wm * 64 + (lane & 31)) * GB_LD + (lane >> 5);
    const int wrow = (wn * 64 + (lane & 31)) * GB_LD + (lane >> 5);
    const int nchunk = (K + GB_K - 1) / GB_K;
    f32x4 ra[2], rw[2];
    auto origin = [&](int t, int &m0, int &n0) {
        const int bid = nwg % 8 == 0 ? (t & 7) * (nwg >> 3) + (t >> 3) : t;
        m0 = (bid / tiles_n) * GB_M; n0 = (bid % tiles_n) * GB_N;
    };
    auto gload = [&](int m0, int n0, int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + lr + 64 * i, col = n0 + lr + 64 * i, k = k0 + kq;
            ra[i] = (row < M && k < K) ? *(const f32x4 *)(A + (size_t)row * lda + k) : f32x4{0, 0, 0, 0};
            rw[i] = (col < N && k < K) ? *(const f32x4 *)(W + (size_t)col * K + k) : f32x4{0, 0, 0, 0};
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float *pa = As[buf] + (lr + 64 * i) * GB_LD + kq, *pw = Ws[buf] + (lr + 64 * i) * GB_LD + kq;
            pa[0] = ra[i].x; pa[1] = ra[i].y; pa[2] = ra[i].z; pa[3] = ra[i].w;
            pw[0] = rw[i].x; pw[1] = rw[i].y; pw[2] = rw[i].z; pw[3] = rw[i].w;
        }
    };
    int m0, n0;
    origin(blockIdx.x, m0, n0);
    gload(m0, n0, 0);
    for (int tile = blockIdx.x; tile < nwg; tile += step) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        sstore(0);
        __syncthreads();
        for (int c = 0; c < nchunk; ++c) {
            const int cur = c & 1;
            if (c + 1 < nchunk) gload(m0, n0, (c + 1) * GB_K);
            const float *as = As[cur], *ws = Ws[cur];
#pragma unroll
            for (int kk = 0; kk < GB_K / 2; ++kk) {
                const float a0 = as[arow + kk * 2], a1 = as[arow + 32 * GB_LD + kk * 2];
                const float b0 = ws[wrow + kk * 2], b1 = ws[wrow + 32 * GB_LD + kk * 2];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (c + 1 < nchunk) sstore(cur ^ 1);
            __syncthreads();
        }
        const int cm0 = m0, cn0 = n0, next = tile + step;
        float bv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = cn0 + wn * 64 + j * 32 + (lane & 31);
            bv[j] = (bias && col < N) ? bias[col] : 0.0f;
        }
        const bool interior = (cm0 + GB_M <= M) && (cn0 + GB_N <= N);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = cn0 + wn * 64 + j * 32 + (lane & 31);
            float *cp = C + (size_t)(cm0 + wm * 64 + 4 * (lane >> 5)) * ldc + col;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = i * 32 + (r & 3) + 8 * (r >> 2);
                    if (interior || (cm0 + wm * 64 + 4 * (lane >> 5) + rr < M && col < N)) cp[(size_t)rr * ldc] = acc[i][j][r] + bv[j];
                }
            if (j == 0 && next < nwg) { origin(next, m0, n0); gload(m0, n0, 0); }
        }
    }
}

