// Development tool: lstm_rec4m_kernel on B1 columns and the two-column kernel on B2 columns, launched together on two streams
// (would a 4-column + 2-column mix fill the tail of a launch whose workgroup count is not a multiple of the CU count?)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I climsim_amd/csrc tools/rec_mix_bench.hip climsim_amd/csrc/rec.hip -o tools/bin/rec_mix_bench
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
void csa_set_error(const char *w, hipError_t e) { fprintf(stderr, "%s: %s\n", w, hipGetErrorString(e)); }
void csa_set_error_msg(const char *m) { fprintf(stderr, "%s\n", m); }
int main(int argc, char **argv)
{
    const int B1 = atoi(argv[1]), B2 = atoi(argv[2]), L = 60, nh = 128, iters = 30, B = B1 + B2;
    std::vector<float> w(4 * nh * nh), wp4(4 * nh * nh), wp2(4 * nh * nh), P((size_t)L * B * 4 * nh), h0((size_t)B * nh);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX - 0.5f); };
    for (auto &x : w) x = 0.17f * rnd();
    for (auto &x : P) x = 2.0f * rnd();
    for (auto &x : h0) x = rnd();
    rec4m_pack_weights(nh, w.data(), wp4.data());
    rec_pack_weights(1, nh, w.data(), wp2.data());
    float *dW4, *dW2, *dP, *dh, *dc, *dH;
    hipMalloc(&dW4, wp4.size() * 4); hipMalloc(&dW2, wp2.size() * 4); hipMalloc(&dP, P.size() * 4); hipMalloc(&dh, h0.size() * 4);
    hipMalloc(&dc, h0.size() * 4); hipMalloc(&dH, (size_t)L * B * nh * 4);
    hipMemcpy(dW4, wp4.data(), wp4.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW2, wp2.data(), wp2.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dh, h0.data(), h0.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, h0.data(), h0.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t e0, e1, j; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&j);
    // the two parts are independent problems here (separate P / H regions of the same buffers)
    float *P2 = dP + (size_t)L * B1 * 4 * nh, *H2 = dH + (size_t)L * B1 * nh;
    auto run = [&] {
        if (B1 > 0) launch_rec4m(nh, dW4, dP, dh, dc, dH, B1, L, 0, s1);
        if (B2 > 0) launch_rec(1, nh, dW2, nullptr, P2, dh, dc, H2, B2, L, 0, s2);
    };
    for (int i = 0; i < 5; ++i) run();
    hipDeviceSynchronize();
    hipEventRecord(e0, s1);
    hipStreamWaitEvent(s2, e0, 0);
    for (int i = 0; i < iters; ++i) run();
    hipEventRecord(j, s2); hipStreamWaitEvent(s1, j, 0);
    hipEventRecord(e1, s1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rec4m %d cols || rec2 %d cols: %.2f us per pair\n", B1, B2, 1e3 * ms / iters);
    return 0;
}
