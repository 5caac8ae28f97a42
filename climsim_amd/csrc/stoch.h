// Internal: the handle of the stochastic recurrent layers, shared by stoch.hip (forward) and stoch_bwd.hip (BPTT).
#pragma once
#include "common.h"
#include <vector>

struct csa_stoch {
    int kind, nx, nh, max_rows;       // kind 0: GRU5, 1: LSTM4
    int has_bias = 0;                 // GRU5 with bias_ih / bias_zh
    float *w_in_t, *b_in;             // (N, nx) transposed input weights for the NT GEMM, optional bias
    float *wp_a, *wp_b, *b_zh;        // packed recurrent rows
    float *XP;
    // training (csa_stoch_enable_training): reference-layout weight copies for the gradient GEMMs, transposed recurrent
    // packings for BPTT, saved activations
    float *w_ref_in = nullptr;        // GRU5: weight_ih (nx,3H); LSTM4: weight_encoder[:nx] (nx,5H)
    float *wT_a = nullptr, *wT_b = nullptr;   // BPTT packings: LSTM4 W_h^T; GRU5 W_zh^T (a), W_enc^T (b)
    float *Hseq = nullptr, *Cseq = nullptr, *ZN = nullptr, *Zs = nullptr, *EX = nullptr, *GZ = nullptr, *GPd = nullptr, *part = nullptr;
    float *own[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // the handle's own XP, Hseq, Cseq, ZN, Zs, EX (csa_stoch_set_activations)
    std::vector<float> host_a, host_b;        // host copies of the recurrent matrices (reference layout) until training is enabled
    std::vector<void *> owned;
};

// host packers (index-map friendly: the trainer runs them on arrays whose values are their own indices)
void stoch_pack_rows(int nh, int R, const float *W, int ncols, int col0, float *packed);     // stoch.hip
void stoch_pack_t(int nh, int ncols, const float *W, float *packed);                          // stoch_bwd.hip
