// Attainable fp32 MFMA rate (development tool): back-to-back independent v_mfma_f32_32x32x2_f32, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x * 1e-9f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(c1 - c0); out[1] = (float)(w1 - w0); }
}
int main(int argc, char **argv)
{
    const int wg_per_cu = argc > 1 ? atoi(argv[1]) : 1, iters = 20000;
    const int grid = 256 * wg_per_cu;
    float *d; hipMalloc(&d, grid * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<grid, 256>>>(d, 100, 1.0f, 1e-6f);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<<<grid, 256>>>(d, iters, 1.0f, 1e-6f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    const double flop = (double)grid * 4 * iters * 16 * 4096.0;
    printf("wg/cu=%d  %.3f ms  %.1f TFLOP/s  block0: %.0f shader cycles / %.0f ticks(100MHz) = %.3f GHz; cycles per MFMA per wave %.1f\n",
           wg_per_cu, ms, flop / (ms * 1e-3) / 1e12, h[0], h[1], h[0] / h[1] * 0.1, h[0] / (iters * 16.0));
    return 0;
}
