// Throughput micro-benchmark of the fp32 issue paths on one CU (development tool):
// v_fma_f32, v_pk_fma_f32 (op_sel broadcast form used by rec.hip), f32 MFMA 4x4x1 / 16x16x4 / 32x32x2,
// at 1, 2, 4 waves per SIMD.  Prints shader cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITERS 2000
template <int MODE>
__global__ void bench(float *out, unsigned long long *cyc)
{
    const int tid = threadIdx.x;
    float a = 1.0f + tid * 1e-3f, b = 0.999f;
    f32x2 acc2[8]; float acc1[16]; f32x4 m4[4]; f32x16 m16[2];
    for (int i = 0; i < 8; ++i) acc2[i] = f32x2{(float)i, (float)i + 1};
    for (int i = 0; i < 16; ++i) acc1[i] = (float)i;
    for (int i = 0; i < 4; ++i) m4[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) m16[i][r] = 0;
    f32x2 w = {a, b}, h = {b, a};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {   // 16 independent v_fma_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc1[i]) : "v"(a), "v"(b));
        } else if (MODE == 1) {   // 16 v_pk_fma_f32 over 8 accumulators
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc2[i & 7]) : "v"(w), "v"(h));
        } else if (MODE == 2) {   // 16 x mfma 4x4x1 (16 blocks)
#pragma unroll
            for (int i = 0; i < 16; ++i) m4[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m4[i & 3], 0, 0, 0);
        } else if (MODE == 3) {   // 16 x mfma 16x16x4
#pragma unroll
            for (int i = 0; i < 16; ++i) m4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m4[i & 3], 0, 0, 0);
        } else if (MODE == 4) {   // 16 x mfma 32x32x2
#pragma unroll
            for (int i = 0; i < 16; ++i) m16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, m16[i & 1], 0, 0, 0);
        } else if (MODE == 5) {   // pk_fma without op_sel modifiers
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i & 7]) : "v"(w), "v"(h));
        } else if (MODE == 6) {   // 8 pk_fma + 8 x mfma 4x4x1 interleaved (do the pipes overlap?)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc2[i & 7]) : "v"(w), "v"(h));
                m4[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m4[i & 3], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc2[i].x + acc2[i].y;
    for (int i = 0; i < 16; ++i) s += acc1[i];
    for (int i = 0; i < 4; ++i) s += m4[i].x + m4[i].y + m4[i].z + m4[i].w;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += m16[i][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
}
// MODE 0: 128 distinct weight pairs, h from registers; MODE 1: + h from LDS (depth-2 compiler schedule)
template <int MODE>
__global__ __launch_bounds__(512, 2) void bench_real(float *out, unsigned long long *cyc, const f32x4 *wsrc)
{
    __shared__ __attribute__((aligned(16))) float hbuf[1088];
    const int tid = threadIdx.x;
    f32x2 w[4][16];
    for (int i = 0; i < 32; ++i) {
        f32x4 v = wsrc[i * 512 + tid];
        w[i / 8][(i % 8) * 2] = f32x2{v.x, v.y};
        w[i / 8][(i % 8) * 2 + 1] = f32x2{v.z, v.w};
    }
    for (int i = tid; i < 1088; i += 512) hbuf[i] = 0.001f * i;
    f32x4 hreg[16];
    for (int j = 0; j < 16; ++j) hreg[j] = f32x4{0.1f * j, 0.2f, 0.3f, 0.4f * tid};
    f32x2 acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS / 8; ++it) {
        const f32x4 *hp = (const f32x4 *)&hbuf[(tid & 3) * 68 + (it & 1) * 544];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            f32x4 hv = MODE == 0 ? hreg[j] : hp[j];
            const f32x2 ha = {hv.x, hv.y}, hb = {hv.z, hv.w};
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[s2]) : "v"(w[s2][j]), "v"(ha));
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[s2]) : "v"(w[s2][j]), "v"(hb));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + tid] = acc[0].x + acc[1].y + acc[2].x + acc[3].y;
    if ((tid & 63) == 0) cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
}
template <int MODE> void run_real(const char *name, float *out, unsigned long long *cyc, const f32x4 *wsrc)
{
    hipLaunchKernelGGL(bench_real<MODE>, dim3(1), dim3(512), 0, 0, out, cyc, wsrc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(16);
    hipMemcpy(c.data(), cyc, 16 * 8, hipMemcpyDeviceToHost);
    printf("%-28s per-wave cycles per 128-pk_fma step:", name);
    for (int i = 0; i < 8; ++i) printf(" %.0f", (double)c[i] / (ITERS / 8));
    printf("\n");
}
template <int MODE> void run(const char *name, float *out, unsigned long long *cyc)
{
    for (int waves : {4, 8, 16}) {
        for (int blocks : {1, 256}) {
            hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc);
            hipDeviceSynchronize();
            std::vector<unsigned long long> c(16);
            hipMemcpy(c.data(), cyc, 16 * 8, hipMemcpyDeviceToHost);
            unsigned long long mx = 0;
            for (int i = 0; i < waves; ++i) mx = c[i] > mx ? c[i] : mx;
            // wave-instructions issued per SIMD = ITERS*16*(waves/4); cycles per wave-instr per SIMD:
            printf("%-28s waves/SIMD %d  blocks %3d : %.2f cycles per wave-instr per SIMD (slowest wave %llu cyc)\n", name,
                   waves / 4, blocks, (double)mx / (ITERS * 16.0 * (waves / 4)), mx);
        }
    }
}
int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    f32x4 *wsrc; hipMalloc(&wsrc, 32 * 512 * 16); hipMemset(wsrc, 0, 32 * 512 * 16);
    run_real<0>("real regs, h in registers", out, cyc, wsrc);
    run_real<1>("real regs, h from LDS", out, cyc, wsrc);
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_pk_fma_f32 op_sel", out, cyc);
    run<5>("v_pk_fma_f32 plain", out, cyc);
    run<2>("mfma_f32_4x4x1_16b", out, cyc);
    run<3>("mfma_f32_16x16x4", out, cyc);
    run<4>("mfma_f32_32x32x2", out, cyc);
    run<6>("pk_fma + mfma4x4x1 pairs", out, cyc);
    return 0;
}
