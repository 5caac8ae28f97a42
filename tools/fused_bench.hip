// Standalone micro-benchmark of the dual-pipe fused LSTM kernel (development tool).
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
void csa_set_error(const char *w, hipError_t e) { fprintf(stderr, "%s: %s\n", w, hipGetErrorString(e)); }
void csa_set_error_msg(const char *m) { fprintf(stderr, "%s\n", m); }
int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 384, L = 60, nh = 128, K = argc > 2 ? atoi(argv[2]) : 144, iters = 100;
    std::vector<float> whh(4 * nh * nh), whhp(4 * nh * nh), wih((size_t)4 * nh * K), wihp((size_t)4 * nh * K), bias(4 * nh),
        X((size_t)L * B * K), h0((size_t)B * nh), c0((size_t)B * nh);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX - 0.5f); };
    for (auto &x : whh) x = 0.17f * rnd();
    for (auto &x : wih) x = 0.17f * rnd();
    for (auto &x : bias) x = rnd();
    for (auto &x : X) x = rnd();
    for (auto &x : h0) x = rnd();
    for (auto &x : c0) x = rnd();
    rec_pack_weights(1, nh, whh.data(), whhp.data());
    fused_pack_wih(nh, K, wih.data(), wihp.data());
    float *dWh, *dWi, *db, *dX, *dh, *dc, *dH;
    hipMalloc(&dWh, whhp.size() * 4); hipMalloc(&dWi, wihp.size() * 4); hipMalloc(&db, bias.size() * 4);
    hipMalloc(&dX, X.size() * 4); hipMalloc(&dh, h0.size() * 4); hipMalloc(&dc, c0.size() * 4);
    hipMalloc(&dH, (size_t)L * B * nh * 4);
    hipMemcpy(dWh, whhp.data(), whhp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dWi, wihp.data(), wihp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dh, h0.data(), h0.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c0.data(), c0.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch_fused_lstm(nh, K, dWh, dWi, db, dX, dh, dc, dH, B, L, 0, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch_fused_lstm(nh, K, dWh, dWi, db, dX, dh, dc, dH, B, L, 0, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = 1e3 * ms / iters;
    printf("fused B=%d K=%d  %.2f us/launch  %.1f ns/step  %.2f TFLOP/s (proj+rec)\n", B, K, us, 1e3 * us / L,
           (double)B * L * 2 * 4 * nh * (nh + K) / (us * 1e-6) / 1e12);
    return 0;
}
