// mlp_api.hip -- the offline Keras MLP baseline:
//   baseline_models/MLP/training/HPO/baseline_v1/step2_retrain/step2_retrain.py:93-121
//   input 124 -> Dense(768,640,512,640,640)+LeakyReLU(0.15) -> Dense(128)+LeakyReLU(0.15)
//         -> Dense(120, linear) || Dense(8, relu)  (concatenated, 128 outputs); 1,753,472 parameters.
// Pure GEMM chain on the fp32 matrix cores: every layer is one launch of proj_gemm_kernel with the bias and
// the activation fused into its epilogue; activations ping-pong between two scratch buffers.
#include "common.h"
#include <vector>

struct csa_mlp {
    int nlayers, max_batch;
    std::vector<int> dims;          // nlayers + 1
    std::vector<float *> W, b;      // device, (out,in) row-major as in PyTorch / transposed Keras kernels
    float alpha;
    int n_lin;
    float *buf[2];
    std::vector<void *> owned;
};

extern "C" int csa_mlp_create(int nlayers, const int *dims, const float *const *weights, const float *const *biases,
                              float leaky_alpha, int n_lin_out, int max_batch, csa_mlp **out)
{
    if (nlayers <= 0 || !dims || !weights || !biases || !out || max_batch <= 0) { csa_set_error_msg("csa_mlp_create: bad argument"); return CSA_ERR_ARG; }
    for (int l = 0; l < nlayers; ++l)
        if (dims[l] % 4) { csa_set_error_msg("csa_mlp_create: layer widths must be multiples of 4"); return CSA_ERR_UNSUPPORTED; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { csa_set_error_msg("csa_mlp_create: no HIP device"); return CSA_ERR_HIP; }
    csa_mlp *h = new csa_mlp();
    h->nlayers = nlayers; h->max_batch = max_batch; h->alpha = leaky_alpha; h->n_lin = n_lin_out;
    h->dims.assign(dims, dims + nlayers + 1);
    int rc = CSA_OK, wmax = 0;
    auto up = [&](const float *src, size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, sizeof(float) * n) != hipSuccess) { rc = CSA_ERR_NOMEM; return (float *)nullptr; }
        h->owned.push_back(p);
        if (src && hipMemcpy(p, src, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
        return (float *)p;
    };
    for (int l = 0; l < nlayers; ++l) {
        h->W.push_back(up(weights[l], (size_t)dims[l + 1] * dims[l]));
        h->b.push_back(up(biases[l], dims[l + 1]));
        wmax = dims[l + 1] > wmax ? dims[l + 1] : wmax;
    }
    h->buf[0] = up(nullptr, (size_t)max_batch * wmax);
    h->buf[1] = up(nullptr, (size_t)max_batch * wmax);
    if (rc != CSA_OK) { for (void *p : h->owned) (void)hipFree(p); delete h; return rc; }
    *out = h;
    return CSA_OK;
}

extern "C" int csa_mlp_destroy(csa_mlp *h)
{
    if (!h) return CSA_ERR_ARG;
    for (void *p : h->owned) (void)hipFree(p);
    delete h;
    return CSA_OK;
}

// x (B, dims[0]) -> y (B, dims[nlayers]); hidden layers LeakyReLU(alpha), last layer split linear|ReLU at n_lin
extern "C" int csa_mlp_forward(csa_mlp *h, int B, const float *x, float *y, void *stream)
{
    if (!h || !x || !y || B <= 0 || B > h->max_batch) { csa_set_error_msg("csa_mlp_forward: bad argument"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const float *in = x;
    for (int l = 0; l < h->nlayers; ++l) {
        const bool last = l + 1 == h->nlayers;
        float *o = last ? y : h->buf[l & 1];
        // up to 1,024 rows the 128x128 tiling cannot fill the part (384 rows x 768 columns = 18 tiles): 32x32 split-K tiles
        int rc = B <= 1024
                     ? launch_gemm_small(in, h->W[l], h->b[l], o, B, h->dims[l + 1], h->dims[l], last ? 2 : 1, h->alpha, last ? h->n_lin : 0, s)
                     : launch_gemm_act(in, h->W[l], h->b[l], o, B, h->dims[l + 1], h->dims[l], last ? 2 : 1, h->alpha,
                                       last ? h->n_lin : 0, s);
        if (rc) return rc;
        in = o;
    }
    return CSA_OK;
}
