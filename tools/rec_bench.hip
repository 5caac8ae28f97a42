// Standalone micro-benchmark of the recurrent kernel (development tool, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I climsim_amd/csrc tools/rec_bench.hip climsim_amd/csrc/rec.hip -o /tmp/rec_bench
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
void csa_set_error(const char *w, hipError_t e) { fprintf(stderr, "%s: %s\n", w, hipGetErrorString(e)); }
void csa_set_error_msg(const char *m) { fprintf(stderr, "%s\n", m); }
int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 384, L = 60, nh = 128, iters = argc > 2 ? atoi(argv[2]) : 50;
    std::vector<float> w(4 * nh * nh), wp(4 * nh * nh), P((size_t)L * B * 4 * nh), h0((size_t)B * nh), c0((size_t)B * nh);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX - 0.5f); };
    for (auto &x : w) x = 0.17f * rnd();
    for (auto &x : P) x = 2.0f * rnd();
    for (auto &x : h0) x = rnd();
    for (auto &x : c0) x = rnd();
    const bool mfma = argc > 3 && argv[3][0] == 'm';       // lstm_rec4m_kernel (matrix pipe; CSA_REC4_KERNEL picks BLGP or not)
    if (mfma) rec4m_pack_weights(nh, w.data(), wp.data());
    else rec_pack_weights(1, nh, w.data(), wp.data());
    float *dW, *dP, *dh, *dc, *dH;
    hipMalloc(&dW, wp.size() * 4); hipMalloc(&dP, P.size() * 4); hipMalloc(&dh, h0.size() * 4);
    hipMalloc(&dc, c0.size() * 4); hipMalloc(&dH, (size_t)L * B * nh * 4);
    hipMemcpy(dW, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dh, h0.data(), h0.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c0.data(), c0.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&] { return mfma ? launch_rec4m(nh, dW, dP, dh, dc, dH, B, L, 0, 0) : launch_rec(1, nh, dW, nullptr, dP, dh, dc, dH, B, L, 0, 0); };
    for (int i = 0; i < 5; ++i) run();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) run();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> H((size_t)L * B * nh);
    hipMemcpy(H.data(), dH, H.size() * 4, hipMemcpyDeviceToHost);
    double cs = 0; for (float x : H) cs += x;
    const double us = 1e3 * ms / iters;
#ifdef REC_EXP_CLOCK
    printf("block0: %.0f shader cycles, %.0f ticks(100MHz) -> %.3f GHz, %.1f cycles/step\n", H[0], H[1], H[0] / H[1] * 0.1, H[0] / L);
#endif
#ifdef REC_EXP_STAMP
    for (int wv = 0; wv < 8; ++wv)
        printf("wave %d: cycles/step  fma(read+fma) %.0f  gates %.0f  write+barrier %.0f\n", wv, H[8 + wv * 4] / L,
               H[9 + wv * 4] / L, H[10 + wv * 4] / L);
#endif
    printf("B=%d  %.2f us/launch  %.1f ns/step  %.2f TFLOP/s  checksum %.6f\n", B, us, 1e3 * us / L,
           (double)B * L * 2 * 4 * nh * nh / (us * 1e-6) / 1e12, cs);
    return 0;
}
