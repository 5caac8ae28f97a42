// pack.h -- host-side weight packers shared by the inference handle (api.hip) and the trainer
// (train_api.hip).  The trainer runs them on arrays whose VALUES are source indices, which turns
// every packer into an index map for the device-side gather that re-packs after an optimiser step.
#pragma once
#include <vector>
#include <cstring>

inline std::vector<float> transposed(const float *w, int O, int K)
{
    std::vector<float> t((size_t)O * K);
    for (int o = 0; o < O; ++o)
        for (int k = 0; k < K; ++k) t[(size_t)k * O + o] = w[(size_t)o * K + k];
    return t;
}

// Input-projection weights with rows permuted to unit-major order n' = u*S + pos.  LSTM: S = 4, pos runs
// over [i, g~, f, o] (the pair order the recurrent kernel's lane groups read); GRU: S = 3, [r, z, n] (round 2: the rows used to
// be padded to 4, i.e. a quarter of the projection GEMM's columns and of P was zeros).
// The bias that can be folded into the projection is b_ih + b_hh (GRU keeps b_hn apart).
inline int gate_stride(int use_lstm) { return use_lstm ? 4 : 3; }
inline void pack_ih(int use_lstm, int nh, int K, const float *w_ih, const float *b_ih, const float *b_hh,
             std::vector<float> &w, std::vector<float> &bias, std::vector<float> &bhn)
{
    const int G = use_lstm ? 4 : 3, S = gate_stride(use_lstm);
    w.assign((size_t)S * nh * K, 0.0f);
    bias.assign((size_t)S * nh, 0.0f);
    bhn.assign((size_t)nh, 0.0f);
    for (int u = 0; u < nh; ++u)
        for (int g = 0; g < G; ++g) {
            static const int lstm_pos[4] = {0, 2, 1, 3};   // PyTorch i,f,g,o -> position in [i,g,f,o]
            const int src = g * nh + u, dst = u * S + (use_lstm ? lstm_pos[g] : g);
            memcpy(&w[(size_t)dst * K], &w_ih[(size_t)src * K], sizeof(float) * K);
            if (!use_lstm && g == 2) { bias[dst] = b_ih[src]; bhn[u] = b_hh[src]; }
            else bias[dst] = b_ih[src] + b_hh[src];
        }
}

