// Standalone micro-benchmark of the projection GEMM (development tool):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I climsim_amd/csrc tools/gemm_bench.hip climsim_amd/csrc/gemm.hip -o tools/bin/gemm_bench
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
void csa_set_error(const char *w, hipError_t e) { fprintf(stderr, "%s: %s\n", w, hipGetErrorString(e)); }
void csa_set_error_msg(const char *m) { fprintf(stderr, "%s\n", m); }
int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 384, K = argc > 2 ? atoi(argv[2]) : 144, N = 512, M = 60 * B, iters = 50;
    std::vector<float> A((size_t)M * K), W((size_t)N * K), bias(N), C((size_t)M * N);
    srand(2);
    auto rnd = [] { return rand() / (float)RAND_MAX - 0.5f; };
    for (auto &x : A) x = rnd();
    for (auto &x : W) x = rnd();
    for (auto &x : bias) x = rnd();
    float *dA, *dW, *db, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&db, N * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bias.data(), N * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch_proj_gemm(dA, dW, db, dC, M, N, K, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch_proj_gemm(dA, dW, db, dC, M, N, K, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 200; ++t) {
        const int r = rand() % M, c = rand() % N;
        double ref = bias[c];
        for (int k = 0; k < K; ++k) ref += (double)A[(size_t)r * K + k] * W[(size_t)c * K + k];
        maxerr = fmax(maxerr, fabs(ref - C[(size_t)r * N + c]));
    }
    const double us = 1e3 * ms / iters;
    printf("M=%d N=%d K=%d  %.2f us  %.1f TFLOP/s  max|err| %.2e\n", M, N, K, us, 2.0 * M * N * K / (us * 1e-6) / 1e12, maxerr);
    return 0;
}
