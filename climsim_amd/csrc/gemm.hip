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
// Global loads for chunk c+1 are issued before the MFMAs of chunk c (register double buffer).
#include "common.h"

#define GB_M 128
#define GB_N 128
#define GB_K 16
#define GB_LD (GB_K + 1)
#define GB_THREADS 256

__global__ __launch_bounds__(GB_THREADS) void proj_gemm_kernel(
    const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
    float *__restrict__ C, int M, int N, int K)
{
    __shared__ float As[GB_M * GB_LD];
    __shared__ float Ws[GB_N * GB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * GB_M, n0 = blockIdx.y * GB_N;

    // staging map: thread -> (row r, r+64; k quad kq)
    const int lr = tid >> 2, kq = (tid & 3) * 4;
    f32x4 ra[2], rw[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + lr + 64 * i, col = n0 + lr + 64 * i, k = k0 + kq;
            ra[i] = (row < M && k < K) ? *(const f32x4 *)(A + (size_t)row * K + k) : f32x4{0, 0, 0, 0};
            rw[i] = (col < N && k < K) ? *(const f32x4 *)(W + (size_t)col * K + k) : f32x4{0, 0, 0, 0};
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float *pa = As + (lr + 64 * i) * GB_LD + kq;
            float *pw = Ws + (lr + 64 * i) * GB_LD + kq;
            pa[0] = ra[i].x; pa[1] = ra[i].y; pa[2] = ra[i].z; pa[3] = ra[i].w;
            pw[0] = rw[i].x; pw[1] = rw[i].y; pw[2] = rw[i].z; pw[3] = rw[i].w;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int arow = (wm * 64 + (lane & 31)) * GB_LD + (lane >> 5);
    const int wrow = (wn * 64 + (lane & 31)) * GB_LD + (lane >> 5);

    gload(0);
    for (int k0 = 0; k0 < K; k0 += GB_K) {
        __syncthreads();            // previous chunk fully consumed
        sstore();
        __syncthreads();
        if (k0 + GB_K < K) gload(k0 + GB_K);
#pragma unroll
        for (int kk = 0; kk < GB_K / 2; ++kk) {
            const float a0 = As[arow + kk * 2], a1 = As[arow + 32 * GB_LD + kk * 2];
            const float b0 = Ws[wrow + kk * 2], b1 = Ws[wrow + 32 * GB_LD + kk * 2];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // epilogue: D[i][j] with j = lane&31 (column) and i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < M) C[(size_t)row * N + col] = acc[i][j][r] + bv;
            }
        }
    }
}

int launch_proj_gemm(const float *A, const float *W, const float *bias, float *C,
                     int M, int N, int K, hipStream_t s)
{
    if (K % 4 != 0) {
        csa_set_error_msg("proj_gemm: K must be a multiple of 4");
        return CSA_ERR_UNSUPPORTED;
    }
    dim3 grid((M + GB_M - 1) / GB_M, (N + GB_N - 1) / GB_N);
    hipLaunchKernelGGL(proj_gemm_kernel, grid, dim3(GB_THREADS), 0, s, A, W, bias, C, M, N, K);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
