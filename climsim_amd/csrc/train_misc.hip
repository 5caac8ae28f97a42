// train_misc.hip -- the non-recurrent kernels of the training step:
//   gemm_tn_partial : C_s(N1,N2) = sum over a slice of rows m of A[m][n1] * Bm[m][n2]   (weight gradients
//                     dW_ih = dP^T X, dW_hh = dP^T H_prev; contraction over the L*B rows, split over workgroups)
//   colsum_partial  : bias gradients  db[n] = sum_m dP[m][n]
//   reduce_partials : deterministic sum of the per-slice partials, scattered (row permutation) and
//                     ACCUMULATED into the canonical flat gradient buffer
//   head_bwd / prep_bwd : backward of head.hip / prep.hip (per-column partial weight gradients)
//   loss_* : huber + energy + water loss of rnn/metrics.py with its analytic gradient
//   adam / gather : optimiser update on the canonical flat parameters, re-packing for the kernels
#include "common.h"
#include "train.h"

// -------------------------------------------------------------------------------------------------
// C_s = A_slice^T * B_slice on the fp32 matrix cores.  128x128 output tile per workgroup, 4 waves
// (2x2) of 2x2 MFMA 32x32x2 tiles; both operands are staged [m][n] exactly as they sit in memory
// (the contraction index m is the slow one), so the MFMA operand reads are lane-contiguous.
#define TN_T 128
#define TN_MK 16
// WM x WN waves of NI x NJ MFMA tiles: (2, 2, 2, 2) = 128 x 128 per workgroup; (4, 1, 1, 5) = 128 x 160, for 128 < N2 <= 160
// (the 144 inputs of rnn1: with 128-wide tiles the second tile column does a full tile of MFMAs for 16 columns -- 187 us
// against 110 for the N2 = 128 contractions of the same step).
template <int WM, int WN, int NI, int NJ>
__global__ __launch_bounds__(256) void gemm_tn_partial_kernel(
    TnSegs segs, int lda, int ldb, float *__restrict__ Cpart, float *__restrict__ Csum,
    int M, int N1, int N2, int rows_per_split, int splits_per_seg, int conv_L, int conv_cin)
{
    static_assert(WM * WN == 4 && WM * NI * 32 == TN_T, "wave layout");
    constexpr int TB = WN * NJ * 32, B4 = TB / 4, NB = (TN_MK * B4 + 255) / 256;      // B tile width; float4 per B row; B loads per thread
    // Csum (optional): partial column sums of A, Csum[split][n1] (the bias gradient rides on the rows this kernel
    // stages anyway; only the blockIdx.y == 0 workgroups keep them).
    // segs: up to TN_MAX_SEGS (A, B) pairs of M rows each, contracted into ONE result (the T_w time steps of a TBPTT
    // window share their weight gradient): split z works on segment z / splits_per_seg.
    const float *__restrict__ A = segs.A[blockIdx.z / splits_per_seg];
    const float *__restrict__ Bm = segs.B[blockIdx.z / splits_per_seg];
    // conv_L > 0: B is the implicit im2col of a kernel-3 'same' convolution input X (ldb = cin): row m is the
    // 3*cin contiguous floats starting at X + (m-1)*ldb, first / last third masked on the first / last level of a
    // column (the weight gradient of Conv1D, cnn_train.hip); N2 = 3*cin.
    __shared__ float As[2][TN_MK][TN_T];     // double-buffered: one barrier per 16-row chunk
    __shared__ float Bs[2][TN_MK][TB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    const int n10 = blockIdx.x * TN_T, n20 = blockIdx.y * TB, split = blockIdx.z;
    const int m_begin = (split % splits_per_seg) * rows_per_split, m_end = min(M, m_begin + rows_per_split);
    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // staging: 16 rows x 128 floats of A = 512 float4, two per thread; 16 rows x TB floats of B.  The global loads of chunk
    // c+1 are issued before the MFMAs of chunk c (register prefetch), so their latency hides behind the MFMA cycles.
    const int sr = tid >> 5, sc = (tid & 31) * 4;
    f32x4 va[2], vb[NB], cs = {0, 0, 0, 0};
    const bool do_cs = Csum != nullptr && blockIdx.y == 0;
    auto gload = [&](int m0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + sr + 8 * h;
            va[h] = f32x4{0, 0, 0, 0};
            if (m < m_end) {
                const float *pa = A + (size_t)m * lda + n10 + sc;
                if (n10 + sc + 3 < N1) va[h] = *(const f32x4 *)pa;
                else for (int e = 0; e < 4; ++e) if (n10 + sc + e < N1) va[h][e] = pa[e];
            }
        }
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const int idx = tid + 256 * h, br = idx / B4, bc = (idx - br * B4) * 4, m = m0 + br;
            vb[h] = f32x4{0, 0, 0, 0};
            if (br < TN_MK && m < m_end) {
                const float *pb = conv_L > 0 ? Bm + ((long)m - 1) * ldb + n20 + bc : Bm + (size_t)m * ldb + n20 + bc;
                bool okb = true;
                if (conv_L > 0) {
                    const int l = m % conv_L, c2 = n20 + bc;
                    okb = !((l == 0 && c2 < conv_cin) || (l == conv_L - 1 && c2 >= 2 * conv_cin));
                }
                if (okb) {
                    if (n20 + bc + 3 < N2) vb[h] = *(const f32x4 *)pb;
                    else for (int e = 0; e < 4; ++e) if (n20 + bc + e < N2) vb[h][e] = pb[e];
                }
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) *(f32x4 *)&As[buf][sr + 8 * h][sc] = va[h];
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const int idx = tid + 256 * h, br = idx / B4, bc = (idx - br * B4) * 4;
            if (br < TN_MK) *(f32x4 *)&Bs[buf][br][bc] = vb[h];
        }
        if (do_cs) cs += va[0] + va[1];      // here, not in gload: the loads are still in flight behind the MFMAs there
    };
    if (m_begin < m_end) {
        gload(m_begin);
        sstore(0);
    }
    __syncthreads();
    int cur = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += TN_MK, cur ^= 1) {
        const bool more = m0 + TN_MK < m_end;
        if (more) gload(m0 + TN_MK);
#pragma unroll
        for (int kk = 0; kk < TN_MK / 2; ++kk) {
            const int kr = kk * 2 + (lane >> 5);
            float a[NI], b[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) a[i] = As[cur][kr][(wm * NI + i) * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = Bs[cur][kr][(wn * NJ + j) * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) sstore(cur ^ 1);
        __syncthreads();
    }
    float *C = Cpart + (size_t)split * N1 * N2;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c2 = n20 + (wn * NJ + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c1 = n10 + (wm * NI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (c1 < N1 && c2 < N2) C[(size_t)c1 * N2 + c2] = acc[i][j][r];
            }
    }
    if (do_cs) {      // block-uniform; the loop's last barrier has released As
        *(f32x4 *)&As[0][sr][sc] = cs;
        __syncthreads();
        if (tid < TN_T && n10 + tid < N1) {
            float t = 0.0f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += As[0][r][tid];
            Csum[(size_t)split * N1 + n10 + tid] = t;
        }
    }
}

int launch_gemm_tn_partial(const float *A, int lda, const float *Bm, int ldb, float *Cpart, int M, int N1, int N2,
                           int nsplit, hipStream_t s)
{
    return launch_gemm_tn_conv(A, lda, Bm, ldb, Cpart, M, N1, N2, nsplit, 0, 0, s);
}

// the same, plus the partial column sums of A into Csum[nsplit][N1]
int launch_gemm_tn_partial_cs(const float *A, int lda, const float *Bm, int ldb, float *Cpart, float *Csum, int M, int N1, int N2,
                              int nsplit, hipStream_t s)
{
    TnSegs g{};
    g.A[0] = A; g.B[0] = Bm; g.n = 1;
    return launch_gemm_tn_segs(g, lda, ldb, Cpart, M, N1, N2, nsplit, 0, 0, s, Csum);
}

int launch_gemm_tn_conv(const float *A, int lda, const float *Bm, int ldb, float *Cpart, int M, int N1, int N2,
                        int nsplit, int conv_L, int conv_cin, hipStream_t s)
{
    TnSegs g{};
    g.A[0] = A; g.B[0] = Bm; g.n = 1;
    return launch_gemm_tn_segs(g, lda, ldb, Cpart, M, N1, N2, nsplit, conv_L, conv_cin, s);
}

// nsplit = TOTAL number of partials (a multiple of segs.n); every segment has M rows
int launch_gemm_tn_segs(const TnSegs &segs, int lda, int ldb, float *Cpart, int M, int N1, int N2,
                        int nsplit, int conv_L, int conv_cin, hipStream_t s, float *Csum)
{
    if (segs.n <= 0 || segs.n > TN_MAX_SEGS || nsplit % segs.n) { csa_set_error_msg("gemm_tn: bad segment count"); return CSA_ERR_ARG; }
    if (conv_L > 0 && (conv_cin % 4 || N2 != 3 * conv_cin)) { csa_set_error_msg("gemm_tn(conv): cin multiple of 4 and N2 = 3*cin required"); return CSA_ERR_UNSUPPORTED; }
    if ((lda % 4) || (ldb % 4)) { csa_set_error_msg("gemm_tn: leading dimensions must be multiples of 4"); return CSA_ERR_UNSUPPORTED; }
    const int sps = nsplit / segs.n;
    int rps = (M + sps - 1) / sps;
    rps = (rps + TN_MK - 1) / TN_MK * TN_MK;
    if (N2 > TN_T && N2 <= 160 && conv_L == 0) {      // one 160-wide tile column instead of two 128-wide ones
        dim3 grid((N1 + TN_T - 1) / TN_T, 1, nsplit);
        hipLaunchKernelGGL((gemm_tn_partial_kernel<4, 1, 1, 5>), grid, dim3(256), 0, s, segs, lda, ldb, Cpart, Csum, M, N1, N2, rps, sps, conv_L, conv_cin);
    } else {
        dim3 grid((N1 + TN_T - 1) / TN_T, (N2 + TN_T - 1) / TN_T, nsplit);
        hipLaunchKernelGGL((gemm_tn_partial_kernel<2, 2, 2, 2>), grid, dim3(256), 0, s, segs, lda, ldb, Cpart, Csum, M, N1, N2, rps, sps, conv_L, conv_cin);
    }
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// partial column sums of a (M, N) matrix: part[s][n].  HBM-bound (reads M*N floats once): a workgroup owns 64 columns
// x one row slice, four row-interleaved thread groups read 256-B row segments, summed through LDS.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ A, float *__restrict__ part, int M, int N, int rps)
{
    __shared__ float sm[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx, s = blockIdx.y;
    const int m0 = s * rps, m1 = min(M, m0 + rps);
    float a0 = 0.0f, a1 = 0.0f;
    if (n < N) {
        int m = m0 + ty;
        for (; m + 4 < m1; m += 8) { a0 += A[(size_t)m * N + n]; a1 += A[(size_t)(m + 4) * N + n]; }
        if (m < m1) a0 += A[(size_t)m * N + n];
    }
    sm[ty][tx] = a0 + a1;
    __syncthreads();
    if (ty == 0 && n < N) part[(size_t)s * N + n] = (sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]);
}
int launch_colsum_partial(const float *A, float *part, int M, int N, int nsplit, hipStream_t s)
{
    const int rps = (M + nsplit - 1) / nsplit;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, nsplit), dim3(256), 0, s, A, part, M, N, rps);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// grad[map ? map[i] : i] += sum_s part[s][i]   (map may send two sources to different targets only;
// idx2, when given, receives the same sum as well: b_ih and b_hh share one gradient)
__device__ __forceinline__ void reduce_partials_body(int i, const float *__restrict__ part, int nsplit, int n,
                                                     const int *__restrict__ map, const int *__restrict__ map2, float *__restrict__ grad)
{
    if (i >= n) return;
    // fixed summation order (s ascending within each of eight interleaved chains, chains combined pairwise):
    // deterministic, and the eight loads per trip are independent (a single dependent chain costs nsplit memory latencies;
    // four chains: 16.4 us for 192 partials of 65,536 floats)
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int s = 0;
    for (; s + 7 < nsplit; s += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += part[(size_t)(s + q) * n + i];
    }
    for (; s < nsplit; ++s) a[0] += part[(size_t)s * n + i];
    const float a_ = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    const int d = map ? map[i] : i;
    if (d >= 0) grad[d] += a_;
    if (map2) { const int d2 = map2[i]; if (d2 >= 0) grad[d2] += a_; }
}
__global__ void reduce_partials_kernel(const float *__restrict__ part, int nsplit, int n,
                                       const int *__restrict__ map, const int *__restrict__ map2,
                                       float *__restrict__ grad)
{
    reduce_partials_body(blockIdx.x * blockDim.x + threadIdx.x, part, nsplit, n, map, map2, grad);
}
// first stage for MANY partials (one per column: head / prep backward): chunk c of `chunks` sums its contiguous run of
// partials into tmp[c][i]; the ordinary reduction then finishes over `chunks`.  Fixed order -> deterministic.
__global__ void reduce_stage1_kernel(const float *__restrict__ part, int nsplit, int n, int per, float *__restrict__ tmp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (i >= n) return;
    const int s0 = c * per, s1 = min(nsplit, s0 + per);
    float a0 = 0.0f, a1 = 0.0f;
    int s = s0;
    for (; s + 1 < s1; s += 2) { a0 += part[(size_t)s * n + i]; a1 += part[(size_t)(s + 1) * n + i]; }
    if (s < s1) a0 += part[(size_t)s * n + i];
    tmp[(size_t)c * n + i] = a0 + a1;
}
int launch_reduce_partials_2stage(const float *part, int nsplit, int n, const int *map, const int *map2, float *grad, float *tmp,
                                  int chunks, hipStream_t s)
{
    if (nsplit <= chunks || !tmp) return launch_reduce_partials(part, nsplit, n, map, map2, grad, s);
    const int per = (nsplit + chunks - 1) / chunks;
    hipLaunchKernelGGL(reduce_stage1_kernel, dim3((n + 255) / 256, chunks), dim3(256), 0, s, part, nsplit, n, per, tmp);
    return launch_reduce_partials(tmp, chunks, n, map, map2, grad, s);
}

// the two-stage form for the queue: the first stage runs now on s, the final reduction over `chunks` joins the queue
int reduce_queue_add_2stage(ReduceJobs &J, const float *part, int nsplit, int n, const int *map, const int *map2, float *tmp, int chunks,
                            hipStream_t s)
{
    if (nsplit <= chunks || !tmp) return reduce_queue_add(J, part, nsplit, n, map, map2);
    const int per = (nsplit + chunks - 1) / chunks;
    hipLaunchKernelGGL(reduce_stage1_kernel, dim3((n + 255) / 256, chunks), dim3(256), 0, s, part, nsplit, n, per, tmp);
    CSA_HIP_CHECK(hipGetLastError());
    return reduce_queue_add(J, tmp, chunks, n, map, map2);
}

// the same reduction for SHORT vectors (biases, head / prep partials: n <= 16 K): with one thread per element the launch is
// a handful of workgroups walking nsplit dependent-latency loads (6 us for 512 elements x 64 partials).  Here eight lanes
// share an element: lane q sums partials q, q+8, ... (two interleaved chains), then three xor-shuffles combine the eight
// sums in a fixed tree -> deterministic, ~3x shorter.
__device__ __forceinline__ void reduce_partials_small_body(int gid, const float *__restrict__ part, int nsplit, int n,
                                                           const int *__restrict__ map, const int *__restrict__ map2, float *__restrict__ grad)
{
    const int i = gid >> 3, q = gid & 7;
    const int ic = i < n ? i : n - 1;                       // all lanes take part in the shuffles
    float a0 = 0.0f, a1 = 0.0f;
    int s = q;
    for (; s + 8 < nsplit; s += 16) {
        a0 += part[(size_t)s * n + ic];
        a1 += part[(size_t)(s + 8) * n + ic];
    }
    if (s < nsplit) a0 += part[(size_t)s * n + ic];
    float a = a0 + a1;
    a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
    if (i >= n || q != 0) return;
    const int d = map ? map[i] : i;
    if (d >= 0) grad[d] += a;
    if (map2) { const int d2 = map2[i]; if (d2 >= 0) grad[d2] += a; }
}
__global__ __launch_bounds__(256) void reduce_partials_small_kernel(const float *__restrict__ part, int nsplit, int n,
                                                                    const int *__restrict__ map, const int *__restrict__ map2,
                                                                    float *__restrict__ grad)
{
    reduce_partials_small_body(blockIdx.x * 256 + threadIdx.x, part, nsplit, n, map, map2, grad);
}

// Several reductions in ONE launch (a backward step queues the reduction of every weight-gradient GEMM and runs them together at
// its end: thirteen launches of 5-11 us, most of them a handful of workgroups, become one).  Each job keeps the scheme
// launch_reduce_partials would have chosen for it, so the sums are bit-identical to separate launches.  The jobs' targets must
// not overlap (different parameter tensors), and every job needs its own partial buffer until the launch.
__global__ __launch_bounds__(256) void reduce_partials_multi_kernel(ReduceJobs J, float *__restrict__ grad)
{
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= J.j[j + 1].blk0) ++j;
    const ReduceJob &r = J.j[j];
    const int gid = ((int)blockIdx.x - r.blk0) * 256 + threadIdx.x;
    if (r.small) reduce_partials_small_body(gid, r.part, r.nsplit, r.n, r.map, r.map2, grad);
    else reduce_partials_body(gid, r.part, r.nsplit, r.n, r.map, r.map2, grad);
}
int reduce_queue_add(ReduceJobs &J, const float *part, int nsplit, int n, const int *map, const int *map2)
{
    if (J.n >= REDUCE_MAX_JOBS) { csa_set_error_msg("reduce_queue_add: queue full"); return CSA_ERR_ARG; }
    ReduceJob &r = J.j[J.n++];
    r.part = part; r.nsplit = nsplit; r.n = n; r.map = map; r.map2 = map2;
    r.small = n <= 16384 && nsplit >= 16;
    r.blk0 = J.nblk;
    J.nblk += r.small ? (8 * n + 255) / 256 : (n + 255) / 256;
    return CSA_OK;
}
int launch_reduce_queue(ReduceJobs &J, float *grad, hipStream_t s)
{
    if (J.n == 0) return CSA_OK;
    hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3(J.nblk), dim3(256), 0, s, J, grad);
    CSA_HIP_CHECK(hipGetLastError());
    J.n = 0; J.nblk = 0;
    return CSA_OK;
}

int launch_reduce_partials(const float *part, int nsplit, int n, const int *map, const int *map2, float *grad, hipStream_t s)
{
    if (n <= 16384 && nsplit >= 16) {
        hipLaunchKernelGGL(reduce_partials_small_kernel, dim3((8 * n + 255) / 256), dim3(256), 0, s, part, nsplit, n, map, map2, grad);
        CSA_HIP_CHECK(hipGetLastError());
        return CSA_OK;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 255) / 256), dim3(256), 0, s, part, nsplit, n, map, map2, grad);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// -------------------------------------------------------------------------------------------------
// head backward, one workgroup per column (current generation, memory model):
//   d_out (B,L,ny) [+ prune mask], d_out_sfc (B,nys), d_mem_out (L,B,nm)  ->  dH2 (L,B,nh2),
//   per-column partial gradients of mlp_output, mlp_latent, mlp_surface_output.
// partial layout per column: [W_out ny*nm | b_out ny | W_lat nm*nh2 | b_lat nm | W_sfo nys*nh2 | b_sfo nys]
__global__ __launch_bounds__(256) void head_bwd_kernel(
    DevModel m, int B, const float *__restrict__ d_out, const float *__restrict__ d_out_sfc,
    const float *__restrict__ d_mem_out, const float *__restrict__ Z, const float *__restrict__ H2,
    float *__restrict__ dH2, float *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = m.cfg.nlev, ny = m.cfg.ny, nys = m.cfg.ny_sfc, nm = m.cfg.nh_mem, nh2 = m.cfg.nh2;
    float *dz = smem;                 // (L, nm)
    float *zs = dz + L * nm;          // (L, nm)
    float *dos = zs + L * nm;         // (L, ny)
    float *hs = dos + L * ny;         // (L, nh2)
    const int b = blockIdx.x, tid = threadIdx.x;
    float *pp = part + (size_t)b * (ny * nm + ny + nm * nh2 + nm + nys * nh2 + nys);
    if (((L * ny) | nm | nh2) & 3) {                 // generic widths: element-wise
        for (int idx = tid; idx < L * ny; idx += 256) dos[idx] = d_out[(size_t)b * L * ny + idx];
        for (int idx = tid; idx < L * nm; idx += 256) zs[idx] = Z[((size_t)(idx / nm) * B + b) * nm + idx % nm];
        for (int idx = tid; idx < L * nh2; idx += 256) hs[idx] = H2[((size_t)(idx / nh2) * B + b) * nh2 + idx % nh2];
    } else {                                          // the column's d_out block is contiguous; Z / H2 rows are B rows apart
        rows_to_lds<2, 256>(dos, 0, d_out + (size_t)b * L * ny, 0, 1, L * ny, tid);
        rows_to_lds<1, 256>(zs, nm, Z + (size_t)b * nm, (size_t)B * nm, L, nm, tid);
        rows_to_lds<8, 256>(hs, nh2, H2 + (size_t)b * nh2, (size_t)B * nh2, L, nh2, tid);
    }
    __syncthreads();
    if (m.cfg.output_prune) {
        for (int idx = tid; idx < 12 * ny; idx += 256) if (idx % ny >= 1) dos[idx] = 0.0f;
        __syncthreads();
    }
    // dz = W_out^T d_out + d_mem_out.  Batches of four levels per thread: the d_mem_out loads of a batch are issued together, and
    // when nm divides 256 a thread keeps one latent channel j, so its W_out column is loaded once.
    {
        constexpr int J = 4;
        const bool fixj = (256 % nm) == 0;
        float w[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) w[v] = m.out_w[min(v, ny - 1) * nm + tid % nm];      // ny <= 8 (launch check); clamped, weight 0 below
        for (int base = 0; base < L * nm; base += 256 * J) {
            float dm[J];
#pragma unroll
            for (int q = 0; q < J; ++q) {
                const int idx = min(base + tid + 256 * q, L * nm - 1), l = idx / nm, j = idx - l * nm;
                dm[q] = d_mem_out ? d_mem_out[((size_t)l * B + b) * nm + j] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < J; ++q) {
                const int idx = base + tid + 256 * q, l = min(idx / nm, L - 1), j = idx - (idx / nm) * nm;
                float a = dm[q];
                if (!fixj) {
#pragma unroll
                    for (int v = 0; v < 8; ++v) w[v] = m.out_w[min(v, ny - 1) * nm + j];
                }
#pragma unroll
                for (int v = 0; v < 8; ++v) a += (v < ny ? dos[l * ny + min(v, ny - 1)] : 0.0f) * w[v];
                if (idx < L * nm) dz[idx] = a;
            }
        }
    }
    __syncthreads();
    // dH2 = W_lat^T dz (+ W_sfo^T d_out_sfc on the last level).  Thread -> hidden unit k is FIXED (k = tid mod nh2, 256 / nh2
    // level groups), so the unit's W_lat column sits in registers for all levels; compile-time trip counts for nm = 16
    // (with run-time bounds hipcc serialises every global load of the inner loop behind an s_waitcnt).
    {
        const int G = 256 / nh2 > 0 ? 256 / nh2 : 1, k = tid % nh2, g = tid / nh2;
        if (nm == 16 && nh2 <= 256) {
            if (g < G) {
                float w[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = m.lat_wt[k * 16 + j];
                for (int l = g; l < L; l += G) {
                    const f32x4 *zr = (const f32x4 *)(dz + l * 16);
                    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 z = zr[q];
                        a0 += z.x * w[4 * q]; a1 += z.y * w[4 * q + 1]; a0 += z.z * w[4 * q + 2]; a1 += z.w * w[4 * q + 3];
                    }
                    float a = a0 + a1;
                    if (l == L - 1)
                        for (int v = 0; v < nys; ++v) a += d_out_sfc[(size_t)b * nys + v] * m.sfo_w[v * nh2 + k];
                    dH2[((size_t)l * B + b) * nh2 + k] = a;
                }
            }
        } else {
            for (int idx = tid; idx < L * nh2; idx += 256) {
                const int l = idx / nh2, kk = idx - l * nh2;
                float a = 0.0f;
                for (int j = 0; j < nm; ++j) a += dz[l * nm + j] * m.lat_wt[kk * nm + j];
                if (l == L - 1)
                    for (int v = 0; v < nys; ++v) a += d_out_sfc[(size_t)b * nys + v] * m.sfo_w[v * nh2 + kk];
                dH2[((size_t)l * B + b) * nh2 + kk] = a;
            }
        }
    }
    // partial weight gradients of this column
    float *p = pp;
    for (int idx = tid; idx < ny * nm; idx += 256) {          // dW_out[v][j] = sum_l d_out[l][v] z[l][j]
        const int v = idx / nm, j = idx - v * nm;
        float a = 0.0f;
        for (int l = 0; l < L; ++l) a += dos[l * ny + v] * zs[l * nm + j];
        p[idx] = a;
    }
    p += ny * nm;
    for (int v = tid; v < ny; v += 256) {
        float a = 0.0f;
        for (int l = 0; l < L; ++l) a += dos[l * ny + v];
        p[v] = a;
    }
    p += ny;
    if (nm == 16 && (nh2 == 128 || nh2 == 256)) {             // dW_lat[j][k] = sum_l dz[l][j] h2[l][k]
        // thread = (hidden unit k, group of 16 * nh2 / 256 latent channels): one h2 read and 1-2 broadcast float4 of dz feed 4-8
        // accumulators per level (the element-wise form below runs 8 x L dependent two-read iterations per thread: 14 us)
        const int k = tid % nh2, g = tid / nh2;
        if (nh2 == 128) {
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
            for (int l = 0; l < L; ++l) {
                const float h = hs[l * 128 + k];
                const f32x4 z0 = *(const f32x4 *)(dz + l * 16 + g * 8), z1 = *(const f32x4 *)(dz + l * 16 + g * 8 + 4);
                a[0] = fmaf(z0.x, h, a[0]); a[1] = fmaf(z0.y, h, a[1]); a[2] = fmaf(z0.z, h, a[2]); a[3] = fmaf(z0.w, h, a[3]);
                a[4] = fmaf(z1.x, h, a[4]); a[5] = fmaf(z1.y, h, a[5]); a[6] = fmaf(z1.z, h, a[6]); a[7] = fmaf(z1.w, h, a[7]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) p[(g * 8 + q) * 128 + k] = a[q];
        } else {
            float a[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = 0.0f;
#pragma unroll 2
            for (int l = 0; l < L; ++l) {
                const float h = hs[l * 256 + k];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 z = *(const f32x4 *)(dz + l * 16 + q4 * 4);
                    a[4 * q4] = fmaf(z.x, h, a[4 * q4]); a[4 * q4 + 1] = fmaf(z.y, h, a[4 * q4 + 1]);
                    a[4 * q4 + 2] = fmaf(z.z, h, a[4 * q4 + 2]); a[4 * q4 + 3] = fmaf(z.w, h, a[4 * q4 + 3]);
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) p[q * 256 + k] = a[q];
        }
    } else {
        for (int idx = tid; idx < nm * nh2; idx += 256) {
            const int j = idx / nh2, k = idx - j * nh2;
            float a = 0.0f;
            for (int l = 0; l < L; ++l) a += dz[l * nm + j] * hs[l * nh2 + k];
            p[idx] = a;
        }
    }
    p += nm * nh2;
    for (int j = tid; j < nm; j += 256) {
        float a = 0.0f;
        for (int l = 0; l < L; ++l) a += dz[l * nm + j];
        p[j] = a;
    }
    p += nm;
    for (int idx = tid; idx < nys * nh2; idx += 256) {        // dW_sfo[v][k] = d_out_sfc[v] h2[L-1][k]
        const int v = idx / nh2, k = idx - v * nh2;
        p[idx] = d_out_sfc[(size_t)b * nys + v] * hs[(L - 1) * nh2 + k];
    }
    p += nys * nh2;
    for (int v = tid; v < nys; v += 256) p[v] = d_out_sfc[(size_t)b * nys + v];
}

int head_bwd_partial_floats(const csa_config &c)
{
    return c.ny * c.nh_mem + c.ny + c.nh_mem * c.nh2 + c.nh_mem + c.ny_sfc * c.nh2 + c.ny_sfc;
}
int launch_head_bwd(const DevModel &m, int B, const float *d_out, const float *d_out_sfc, const float *d_mem_out,
                    const float *Z, const float *H2, float *dH2, float *part, hipStream_t s)
{
    const csa_config &c = m.cfg;
    if (c.nh_mem <= 0) { csa_set_error_msg("head_bwd: memory model required"); return CSA_ERR_UNSUPPORTED; }
    if (c.ny > 8) { csa_set_error_msg("head_bwd: at most 8 level outputs"); return CSA_ERR_UNSUPPORTED; }
    const size_t shm = sizeof(float) * (size_t)c.nlev * (2 * c.nh_mem + c.ny + c.nh2);
    hipLaunchKernelGGL(head_bwd_kernel, dim3(B), dim3(256), shm, s, m, B, d_out, d_out_sfc, d_mem_out, Z, H2, dH2, part);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// -------------------------------------------------------------------------------------------------
// prep backward, one workgroup per column:  dX1 (L,B,nh1+nm) sequence order  ->
//   d_mem_in (L,B,nm) level order;  partial grads of mlp_initial (through tanh), mlp_surface1/2 (from
//   d h0/c0 of rnn1), mlp_toa1/2 (from d h0/c0 of rnn2).
// partial layout: [W_init nh1*nxp | b_init nh1 | W_s1 nh1*nxs | b_s1 nh1 | W_s2 nh1*nxs | b_s2 nh1 |
//                  W_toa1 nh2*2 | b_toa1 nh2 | W_toa2 nh2*2 | b_toa2 nh2]
#define PB_U 30            // levels per batch of independent global loads (12: 36.1 us per launch)
__global__ __launch_bounds__(128) void prep_bwd_kernel(
    DevModel m, int B, const float *__restrict__ dX1, const float *__restrict__ X1, const float *__restrict__ X16,
    const float *__restrict__ xs_n, const float *__restrict__ hc0, const float *__restrict__ dhc1,
    const float *__restrict__ dhc2, float *__restrict__ d_mem_in, float *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = m.cfg.nlev, nxp = m.cfg.nx + 1, nxs = m.cfg.nx_sfc, nh1 = m.cfg.nh1, nh2 = m.cfg.nh2;
    const int nm = m.cfg.nh_mem, nin1 = nh1 + nm, nhm = nh1 > nh2 ? nh1 : nh2;
    // (L, 32) level order, rows ZERO-PADDED to 32: the accumulation loop below then has a compile-time trip count and
    // no branch per element.  With "if (v < nxp)" inside it hipcc emitted a scalar branch + s_waitcnt per LDS read and
    // serialised the two global loads of every level behind them: 80 us per launch, 3 launches per training step.
    float *x16 = smem;
    float *xs = x16 + L * 32;          // (nxs)
    const int b = blockIdx.x, tid = threadIdx.x;
    if (nxp & 3) {
        for (int idx = tid; idx < L * 32; idx += 128) {
            const int l = idx >> 5, v = idx & 31;
            x16[idx] = v < nxp ? X16[((size_t)b * L + l) * nxp + v] : 0.0f;
        }
    } else {                                          // batched float4 loads of the column's (L, nxp) block, then the zero padding
        rows_to_lds<2, 128>(x16, 32, X16 + (size_t)b * L * nxp, nxp, L, nxp, tid);
        for (int idx = tid; idx < L * 32; idx += 128) if ((idx & 31) >= nxp) x16[idx] = 0.0f;
    }
    for (int v = tid; v < nxs; v += 128) xs[v] = xs_n[(size_t)b * nxs + v];
    __syncthreads();
    float *p = part + (size_t)b * prep_bwd_partial_floats(m.cfg);
    // mlp_initial: a = tanh(.) saved in X1; dA = dX1 * (1 - a^2)
    for (int j = tid; j < nh1; j += 128) {
        float gw[32];
#pragma unroll
        for (int v = 0; v < 32; ++v) gw[v] = 0.0f;
        float gb = 0.0f;
        for (int t0 = 0; t0 < L; t0 += PB_U) {
            float a[PB_U], dx[PB_U];
#pragma unroll
            for (int u = 0; u < PB_U; ++u) {           // 2 * PB_U independent loads in flight (clamped index on the tail, weight 0)
                const size_t r = ((size_t)min(t0 + u, L - 1) * B + b) * nin1 + j;
                a[u] = X1[r];
                dx[u] = dX1[r];
            }
#pragma unroll
            for (int u = 0; u < PB_U; ++u) {
                const float dA = t0 + u < L ? dx[u] * (1.0f - a[u] * a[u]) : 0.0f;
                // X1 row t is level L-1-t (rnn1 runs upward over the flipped sequence) or, for the stochastic variant whose
                // first RNN runs downward, level t
                const int tt = min(t0 + u, L - 1);
                const f32x4 *xr = (const f32x4 *)(x16 + (m.cfg.add_stochastic_layer ? tt : L - 1 - tt) * 32);
                gb += dA;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 xv = xr[q];
                    gw[4 * q] += dA * xv.x; gw[4 * q + 1] += dA * xv.y; gw[4 * q + 2] += dA * xv.z; gw[4 * q + 3] += dA * xv.w;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 32; ++v) if (v < nxp) p[j * nxp + v] = gw[v];
        p[nh1 * nxp + j] = gb;
    }
    p += nh1 * nxp + nh1;
    // surface MLPs: h0 = tanh(W1 xs + b1) (saved in hc0 slot 0), c0 = W2 xs + b2
    for (int j = tid; j < nh1; j += 128) {
        const float h0 = hc0[((size_t)0 * B + b) * nhm + j];
        const float d1 = dhc1[((size_t)0 * B + b) * nhm + j] * (1.0f - h0 * h0);
        const float d2 = dhc1[((size_t)1 * B + b) * nhm + j];
        for (int v = 0; v < nxs; ++v) {
            p[j * nxs + v] = d1 * xs[v];
            p[nh1 * nxs + nh1 + j * nxs + v] = d2 * xs[v];
        }
        p[nh1 * nxs + j] = d1;
        p[2 * nh1 * nxs + nh1 + j] = d2;
    }
    p += 2 * (nh1 * nxs + nh1);
    {
        const float t0 = xs[1], t1 = xs[6];
        for (int j = tid; j < nh2; j += 128) {
            const float d1 = dhc2[((size_t)0 * B + b) * nhm + j], d2 = dhc2[((size_t)1 * B + b) * nhm + j];
            p[j * 2] = d1 * t0; p[j * 2 + 1] = d1 * t1; p[nh2 * 2 + j] = d1;
            p[nh2 * 3 + j * 2] = d2 * t0; p[nh2 * 3 + j * 2 + 1] = d2 * t1; p[nh2 * 5 + j] = d2;
        }
    }
    // gradient w.r.t. the incoming memory (level order)
    if (d_mem_in) {
        constexpr int J = 8;                          // eight independent loads in flight per thread (a rolled copy loop: one round trip per trip)
        for (int base = 0; base < L * nm; base += 128 * J) {
            float v[J];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int idx = min(base + tid + 128 * j, L * nm - 1), t = idx / nm, k = idx - t * nm;
                v[j] = dX1[((size_t)t * B + b) * nin1 + nh1 + k];
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int idx = base + tid + 128 * j, t = idx / nm, k = idx - t * nm;
                if (idx < L * nm) d_mem_in[((size_t)(m.cfg.add_stochastic_layer ? t : L - 1 - t) * B + b) * nm + k] = v[j];
            }
        }
    }
}

int launch_prep_bwd(const DevModel &m, int B, const float *dX1, const float *X1, const float *X16, const float *xs_n,
                    const float *hc0, const float *dhc1, const float *dhc2, float *d_mem_in, float *part, hipStream_t s)
{
    if (m.cfg.nx + 1 > 32) { csa_set_error_msg("prep_bwd: at most 31 level inputs"); return CSA_ERR_UNSUPPORTED; }
    const size_t shm = sizeof(float) * ((size_t)m.cfg.nlev * 32 + m.cfg.nx_sfc);
    hipLaunchKernelGGL(prep_bwd_kernel, dim3(B), dim3(128), shm, s, m, B, dX1, X1, X16, xs_n, hc0, dhc1, dhc2, d_mem_in, part);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// -------------------------------------------------------------------------------------------------
// Loss of the reference trainer (rnn/utils.py:1203-1366 with rnn/metrics.py:142-163,193-315):
//   loss = huber(pred, target) + w_h * energy_mse + w_w * water_mse
// N = T_w*B samples ordered (tau, b).  Pass 1 (one workgroup per sample): huber partial sum and
// the column integrals e_pred, e_true, w_pred, w_true.  Pass 2 (one workgroup): scalars and the
// per-column window means.  Pass 3 (per sample): analytic gradient w.r.t. pred / pred_sfc.
struct LossConsts { float cp, Lv, Ls, ginv_e, ginv_w; };
__device__ __forceinline__ LossConsts loss_consts() { return {1004.0f, 2.5104e6f, 2.8440e6f, 0.1020408163f, 0.1019716213f}; }

__device__ __forceinline__ void mp_post(float T, float ql, float qi, float dT, float dqn, float &dql, float &dqi, float &lf, float &qn_new, bool &inside)
{
    const float T_new = T + dT * 1200.0f;
    const float raw = (T_new - 253.16f) * 0.05f;
    inside = raw > 0.0f && raw < 1.0f;
    lf = fminf(fmaxf(raw, 0.0f), 1.0f);
    qn_new = (ql + qi) + dqn * 1200.0f;
    dql = (lf * qn_new - ql) * 0.0008333333333333334f;
    dqi = ((1.0f - lf) * qn_new - qi) * 0.0008333333333333334f;
}

// mp_mode -1 (models.py:303-329): the liquid fraction is the model's 4th output itself (de-normalised; the clamped diagnosed value
// is overwritten, :319), so d lf / d dT = 0 and d(dql, dqi) / d lf = +- qn_new / 1200
__device__ __forceinline__ void mp_post_pred(float ql, float qi, float dqn, float lf, float &dql, float &dqi, float &qn_new)
{
    qn_new = (ql + qi) + dqn * 1200.0f;
    dql = (lf * qn_new - ql) * 0.0008333333333333334f;
    dqi = ((1.0f - lf) * qn_new - qi) * 0.0008333333333333334f;
}

// mp_mode -2 (models.py:286-301), ahead of the liquid / ice partition of mode -1: the model predicts the total-water tendency
// (column 1) and f = (cloud fraction of total water)^(1/4) (column 2); c = clamp(f^4, 0, 1),
//   q_tot' = q_n + q_v + 1200 dq_tot,  dq_v = ((1 - c) q_tot' - q_v) / 1200,  dq_n = (c q_tot' - q_n) / 1200
// with q_v the LAST input column.  dcdf = dc/df (0 where the clamp is active).
__device__ __forceinline__ void mp_total_water(float qv, float qn, float dqtot, float f, float &dqv, float &dqn, float &c, float &dcdf, float &qtot_new)
{
    const float f2 = f * f, c_raw = f2 * f2;
    c = fminf(fmaxf(c_raw, 0.0f), 1.0f);
    dcdf = (c_raw > 0.0f && c_raw < 1.0f) ? 4.0f * f2 * f : 0.0f;
    qtot_new = (qn + qv) + dqtot * 1200.0f;
    dqv = ((1.0f - c) * qtot_new - qv) * 0.0008333333333333334f;
    dqn = (c * qtot_new - qn) * 0.0008333333333333334f;
}

__device__ __forceinline__ float block_sum(float v, float *red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.0f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// per-sample scalars: [huber_sum, sq_sum, abs_sum, e_pred, e_true, w_pred, w_true, prec_pred, prec_true]
#define LOSS_NS 9
__global__ __launch_bounds__(128) void loss_pass1_kernel(
    DevModel m, const float *__restrict__ hyai, const float *__restrict__ hybi, int N,
    const float *__restrict__ pred, const float *__restrict__ pred_sfc, const float *__restrict__ tgt,
    const float *__restrict__ tgt_sfc, const float *__restrict__ yto, const float *__restrict__ yto_sfc,
    const float *__restrict__ x_raw, const float *__restrict__ sp, float *__restrict__ samp)
{
    __shared__ float red[4];
    const int L = m.cfg.nlev, ny = m.cfg.ny, nys = m.cfg.ny_sfc, nx = m.cfg.nx;
    const int n = blockIdx.x, tid = threadIdx.x;
    const LossConsts k = loss_consts();
    float hub = 0.f, sq = 0.f, ab = 0.f, ep = 0.f, et = 0.f, wp = 0.f, wt = 0.f;
    const int ne = L * ny + nys;
    for (int i = tid; i < ne; i += 128) {
        const float d = i < L * ny ? pred[(size_t)n * L * ny + i] - tgt[(size_t)n * L * ny + i]
                                   : pred_sfc[(size_t)n * nys + i - L * ny] - tgt_sfc[(size_t)n * nys + i - L * ny];
        const float ad = fabsf(d);
        hub += ad < 1.0f ? 0.5f * d * d : ad - 0.5f;
        sq += d * d; ab += ad;
    }
    const float spn = sp[n];
    for (int l = tid; l < L; l += 128) {
        const float *o = pred + ((size_t)n * L + l) * ny, *ys = m.yscale_lev + l * ny;
        const float *xr = x_raw + ((size_t)n * L + l) * nx, *yt = yto + ((size_t)n * L + l) * 6;
        float dql, dqi, lf, qn; bool in;
        const float dT = o[0] / ys[0];
        float dqv = o[1] / ys[1], dqn = o[2] / ys[2];
        if (m.cfg.mp_mode == -2) { float c, dcdf, qt; mp_total_water(xr[nx - 1], xr[2] + xr[3], o[1] / ys[1], o[2] / ys[2], dqv, dqn, c, dcdf, qt); }
        if (m.cfg.mp_mode == 1) mp_post(xr[0], xr[2], xr[3], dT, dqn, dql, dqi, lf, qn, in);
        else mp_post_pred(xr[2], xr[3], dqn, o[3] / ys[3], dql, dqi, qn);
        const float dhyb = hybi[l + 1] - hybi[l], dhya = hyai[l + 1] - hyai[l];
        const float th_e = k.ginv_e * (spn * dhyb + 100000.0f * dhya);
        const float th_w = k.ginv_w * (spn * dhyb + 100000.0f * dhya);
        ep += th_e * (dT * k.cp - dql * k.Lv - dqi * k.Ls);
        et += th_e * (yt[0] * k.cp - yt[2] * k.Lv - yt[3] * k.Ls);
        wp += th_w * ((dqv + dql) + dqi);
        wt += th_w * ((yt[1] + yt[2]) + yt[3]);
    }
    hub = block_sum(hub, red); sq = block_sum(sq, red); ab = block_sum(ab, red);
    ep = block_sum(ep, red); et = block_sum(et, red); wp = block_sum(wp, red); wt = block_sum(wt, red);
    if (tid == 0) {
        const float snow_p = 1000.0f * (pred_sfc[(size_t)n * nys + 2] / m.yscale_sca[2]);
        const float prec_p = 1000.0f * (pred_sfc[(size_t)n * nys + 3] / m.yscale_sca[3]);
        const float snow_t = 1000.0f * yto_sfc[(size_t)n * nys + 2], prec_t = 1000.0f * yto_sfc[(size_t)n * nys + 3];
        float *s = samp + (size_t)n * LOSS_NS;
        s[0] = hub; s[1] = sq; s[2] = ab;
        s[3] = ep - (prec_p - snow_p) * k.Lv - snow_p * k.Ls;
        s[4] = et - (prec_t - snow_t) * k.Lv - snow_t * k.Ls;
        s[5] = wp + prec_p;            // lhs - rhs, rhs = -precip
        s[6] = wt + prec_t;
        s[7] = prec_p * 0.001f; s[8] = prec_t * 0.001f;
    }
}

// scalars out: [loss, huber, mse, mae, energy, water, precip_sum_mse];  colcoef (B): d energy_mse / d e_pred per sample
__global__ __launch_bounds__(256) void loss_pass2_kernel(
    DevModel m, int B, int Tw, float w_h, float w_w, const float *__restrict__ samp,
    float *__restrict__ scal, float *__restrict__ ecoef)
{
    __shared__ float red[4];
    const int L = m.cfg.nlev, ny = m.cfg.ny, nys = m.cfg.ny_sfc, N = B * Tw, tid = threadIdx.x;
    float hub = 0.f, sq = 0.f, ab = 0.f, wat = 0.f;
    for (int n = tid; n < N; n += 256) {
        const float *s = samp + (size_t)n * LOSS_NS;
        hub += s[0]; sq += s[1]; ab += s[2];
        const float d = s[5] - s[6];
        wat += d * d;
    }
    float en = 0.f, pr = 0.f;
    for (int b = tid; b < B; b += 256) {
        float ep = 0.f, et = 0.f, pp = 0.f, pt = 0.f;
        for (int t = 0; t < Tw; ++t) {
            const float *s = samp + (size_t)(t * B + b) * LOSS_NS;
            ep += s[3]; et += s[4]; pp += s[7]; pt += s[8];
        }
        const float d = ep / Tw - et / Tw;
        en += d * d;
        ecoef[b] = 2.0f * d / ((float)B * (float)Tw);
        pr += (pt - pp) * (pt - pp);
    }
    hub = block_sum(hub, red); sq = block_sum(sq, red); ab = block_sum(ab, red); wat = block_sum(wat, red);
    en = block_sum(en, red); pr = block_sum(pr, red);
    if (tid == 0) {
        const float ntot = (float)N * (float)(L * ny + nys);
        scal[1] = hub / ntot; scal[2] = sq / ntot; scal[3] = ab / ntot;
        scal[4] = en / B; scal[5] = wat / N; scal[6] = pr / B / ((float)Tw * (float)Tw);
        scal[0] = scal[1] + w_h * scal[4] + w_w * scal[5];
    }
}

__global__ __launch_bounds__(128) void loss_pass3_kernel(
    DevModel m, const float *__restrict__ hyai, const float *__restrict__ hybi, int B, int Tw, float w_h, float w_w,
    const float *__restrict__ pred, const float *__restrict__ pred_sfc, const float *__restrict__ tgt,
    const float *__restrict__ tgt_sfc, const float *__restrict__ x_raw, const float *__restrict__ sp,
    const float *__restrict__ samp, const float *__restrict__ ecoef, float *__restrict__ d_pred, float *__restrict__ d_pred_sfc)
{
    const int L = m.cfg.nlev, ny = m.cfg.ny, nys = m.cfg.ny_sfc, nx = m.cfg.nx, N = B * Tw;
    const int n = blockIdx.x, tid = threadIdx.x, b = n % B;
    const LossConsts k = loss_consts();
    const float inv_ntot = 1.0f / ((float)N * (float)(L * ny + nys));
    const float ce = w_h * ecoef[b];                                           // d loss / d e_pred[n]
    const float cw = w_w * 2.0f * (samp[(size_t)n * LOSS_NS + 5] - samp[(size_t)n * LOSS_NS + 6]) / (float)N;  // d loss / d wdiff_pred[n]
    const float spn = sp[n];
    for (int l = tid; l < L; l += 128) {
        const float *o = pred + ((size_t)n * L + l) * ny, *tg = tgt + ((size_t)n * L + l) * ny;
        const float *ys = m.yscale_lev + l * ny, *xr = x_raw + ((size_t)n * L + l) * nx;
        float g[8];
        for (int v = 0; v < ny; ++v) {
            const float d = o[v] - tg[v];
            g[v] = fminf(fmaxf(d, -1.0f), 1.0f) * inv_ntot;
        }
        float dql, dqi, lf, qn; bool in = false;
        const bool diag = m.cfg.mp_mode == 1, total = m.cfg.mp_mode == -2;
        float dqv_, dqn_ = o[2] / ys[2], cfrac = 0.0f, dcdf = 0.0f, qtot_new = 0.0f;
        if (total) mp_total_water(xr[nx - 1], xr[2] + xr[3], o[1] / ys[1], o[2] / ys[2], dqv_, dqn_, cfrac, dcdf, qtot_new);
        if (diag) mp_post(xr[0], xr[2], xr[3], o[0] / ys[0], dqn_, dql, dqi, lf, qn, in);
        else { lf = o[3] / ys[3]; mp_post_pred(xr[2], xr[3], dqn_, lf, dql, dqi, qn); }
        const float dhyb = hybi[l + 1] - hybi[l], dhya = hyai[l + 1] - hyai[l];
        const float th_e = k.ginv_e * (spn * dhyb + 100000.0f * dhya);
        const float th_w = k.ginv_w * (spn * dhyb + 100000.0f * dhya);
        // gradients w.r.t. the post-processed tendencies
        const float g_dT = ce * th_e * k.cp;
        const float g_dql = -ce * th_e * k.Lv + cw * th_w;
        const float g_dqi = -ce * th_e * k.Ls + cw * th_w;
        const float g_dqv = cw * th_w;
        // chain through the microphysics partition: dql = (lf qn_new - ql)/1200, dqi = ((1-lf) qn_new - qi)/1200
        const float dlf = in ? 0.05f * 1200.0f : 0.0f;                          // d lf / d dT
        const float s1200 = 0.0008333333333333334f;
        const float ddT = g_dT + (g_dql - g_dqi) * dlf * qn * s1200;
        const float ddqn = (g_dql * lf + g_dqi * (1.0f - lf)) * 1200.0f * s1200;
        g[0] += ddT / ys[0];
        if (total) {     // through dq_v = ((1-c) q_tot' - q_v)/1200, dq_n = (c q_tot' - q_n)/1200, c = clamp(f^4)
            g[1] += (g_dqv * (1.0f - cfrac) + ddqn * cfrac) / ys[1];
            g[2] += (ddqn - g_dqv) * qtot_new * s1200 * dcdf / ys[2];
        } else {
            g[1] += g_dqv / ys[1];
            g[2] += ddqn / ys[2];
        }
        if (!diag) g[3] += (g_dql - g_dqi) * qn * s1200 / ys[3];                   // through the predicted liquid fraction
        float *dp = d_pred + ((size_t)n * L + l) * ny;
        for (int v = 0; v < ny; ++v) dp[v] = g[v];
    }
    for (int v = tid; v < nys; v += 128) {
        const float d = pred_sfc[(size_t)n * nys + v] - tgt_sfc[(size_t)n * nys + v];
        float g = fminf(fmaxf(d, -1.0f), 1.0f) * inv_ntot;
        // energy: - rain*Lv - snow*Ls with snow = 1000*sfc2, prec = 1000*sfc3; water: + prec
        if (v == 2) g += ce * (1000.0f * (k.Lv - k.Ls)) / m.yscale_sca[2];
        if (v == 3) g += (ce * (-1000.0f * k.Lv) + cw * 1000.0f) / m.yscale_sca[3];
        d_pred_sfc[(size_t)n * nys + v] = g;
    }
}

int launch_loss(const DevModel &m, const float *hyai, const float *hybi, int B, int Tw, float w_h, float w_w,
                const float *pred, const float *pred_sfc, const float *tgt, const float *tgt_sfc, const float *yto,
                const float *yto_sfc, const float *x_raw, const float *sp, float *samp, float *ecoef, float *scal,
                float *d_pred, float *d_pred_sfc, hipStream_t s)
{
    if ((m.cfg.mp_mode != 1 && m.cfg.mp_mode != -1 && m.cfg.mp_mode != -2) || m.cfg.ny > 8) { csa_set_error_msg("loss: mp_mode 1, -1 and -2"); return CSA_ERR_UNSUPPORTED; }
    const int N = B * Tw;
    hipLaunchKernelGGL(loss_pass1_kernel, dim3(N), dim3(128), 0, s, m, hyai, hybi, N, pred, pred_sfc, tgt, tgt_sfc, yto, yto_sfc, x_raw, sp, samp);
    hipLaunchKernelGGL(loss_pass2_kernel, dim3(1), dim3(256), 0, s, m, B, Tw, w_h, w_w, samp, scal, ecoef);
    if (d_pred)
        hipLaunchKernelGGL(loss_pass3_kernel, dim3(N), dim3(128), 0, s, m, hyai, hybi, B, Tw, w_h, w_w, pred, pred_sfc, tgt, tgt_sfc, x_raw, sp, samp, ecoef, d_pred, d_pred_sfc);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// -------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m1,
                            float *__restrict__ m2, int n, float lr, float b1, float b2, float eps, float bc1, float bc2,
                            float wd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // torch.optim.Adam (no amsgrad): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    // p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
    float gi = g[i] + wd * p[i];
    const float a = b1 * m1[i] + (1.0f - b1) * gi;
    const float v = b2 * m2[i] + (1.0f - b2) * gi * gi;
    m1[i] = a; m2[i] = v;
    p[i] -= (lr / bc1) * a / (sqrtf(v) / sqrtf(bc2) + eps);
}
int launch_adam(float *p, const float *g, float *m1, float *m2, int n, float lr, float b1, float b2, float eps, int step,
                float wd, hipStream_t s)
{
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, g, m1, m2, n, lr, b1, b2, eps, bc1, bc2, wd);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// dst[i] = idx[i] >= 0 ? src[idx[i]] (+ src[idx2[i]] if idx2) : 0     (re-packing canonical -> kernel layouts)
__global__ void gather_kernel(float *__restrict__ dst, const float *__restrict__ src, const int *__restrict__ idx,
                              const int *__restrict__ idx2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = idx[i];
    float v = a >= 0 ? src[a] : 0.0f;
    if (idx2) { const int c = idx2[i]; if (c >= 0) v += src[c]; }
    dst[i] = v;
}
// all re-packings of an optimiser step in ONE launch: blockIdx.y selects the table entry (27 separate 4-us launches before)
__global__ void gather_multi_kernel(const GatherEntry *__restrict__ tab, const float *__restrict__ src)
{
    const GatherEntry g = tab[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    const int a = g.idx[i];
    float v = a >= 0 ? src[a] : 0.0f;
    if (g.idx2) { const int c = g.idx2[i]; if (c >= 0) v += src[c]; }
    g.dst[i] = v;
}
int launch_gather_multi(const GatherEntry *tab_dev, int count, int max_n, const float *src, hipStream_t s)
{
    hipLaunchKernelGGL(gather_multi_kernel, dim3((max_n + 255) / 256, count), dim3(256), 0, s, tab_dev, src);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
int launch_gather(float *dst, const float *src, const int *idx, const int *idx2, int n, hipStream_t s)
{
    hipLaunchKernelGGL(gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, idx, idx2, n);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
