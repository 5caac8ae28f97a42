// Can the packed-FMA vector pipe and the fp32 matrix pipe of a SIMD run at full rate at the same time from different
// waves?  (development tool)  One workgroup per CU of 12 waves: waves 0-7 issue independent v_pk_fma_f32 (two per SIMD),
// waves 8-11 issue v_mfma_f32_32x32x2_f32 (one per SIMD).  mode 1 = VALU waves only, 2 = MFMA waves only, 3 = both.
//   hipcc --offload-arch=gfx950 -O3 tools/coexec_bench.hip -o tools/bin/coexec_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PK_FMA(acc, w, h) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(h))
__global__ __launch_bounds__(768) void k(float *out, int iters, int mode, float a, float b)
{
    const int wave = threadIdx.x >> 6;
    const unsigned long long c0 = __builtin_readcyclecounter();
    float res = 0.f;
    if (wave < 8) {
        if (mode & 1) {
            f32x2 acc[8], w = {a, b}, h = {b, a};
            for (int i = 0; i < 8; ++i) acc[i] = f32x2{(float)threadIdx.x, 1.f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) PK_FMA(acc[i], w, h);       // 128 packed FMAs per iteration = one recurrence step
            }
            for (int i = 0; i < 8; ++i) res += acc[i].x + acc[i].y;
        }
    } else if (mode & 2) {
        f32x16 acc[2];
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x * 1e-9f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 9; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);   // 18 MFMAs per iteration
        }
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * 768 + threadIdx.x] = res;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = (float)(c1 - c0);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 768 * 4);
    const int iters = 2000;
    for (int mode = 1; mode <= 3; ++mode) {
        k<<<256, 768>>>(d, 10, mode, 1.0f, 1e-6f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); k<<<256, 768>>>(d, iters, mode, 1.0f, 1e-6f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float h[12]; hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
        printf("mode %d: %.3f ms | cycles per iteration: VALU wave0 %.0f wave4 %.0f (128 pk_fma each; 2 waves/SIMD) | MFMA wave8 %.0f (18 MFMA = 1152 matrix cycles)\n",
               mode, ms, h[0] / iters, h[4] / iters, h[8] / iters);
    }
    return 0;
}
