// mlp_train.hip -- one optimiser step of the Keras MLP baseline (training side):
//   baseline_models/MLP/training/HPO/baseline_v1/step2_retrain/step2_retrain.py:93-155  Dense + LeakyReLU(0.15) stack, split
//   Dense(120, linear) || Dense(8, relu) output, loss 'mse', keras.optimizers.Adam.
// Dense chain on the fp32 matrix cores: forward keeps every layer output; backward per layer = split-M TN GEMM for dW,
// HBM-bound column sums for db, NT GEMM against the transposed weight for the input gradient with the LeakyReLU
// derivative (from the saved OUTPUT: its sign is the pre-activation's) fused into the epilogue.  One flat parameter /
// gradient buffer in state order [W_0 (out,in) | b_0 | W_1 | ...] -> one all-reduce per data-parallel step.
#include "common.h"
#include "train.h"
#include <vector>

struct csa_mlp_trainer {
    int nlayers, max_batch, n_lin;
    float alpha;
    std::vector<int> dims;
    std::vector<size_t> w_off, b_off;
    size_t n_params = 0;
    float *params = nullptr, *m1 = nullptr, *m2 = nullptr, *wT = nullptr, *part = nullptr, *lpart = nullptr;
    std::vector<float *> act;      // act[l]: output of layer l (B, dims[l+1])
    float *g0 = nullptr, *g1 = nullptr;
    const float *x_in = nullptr;
    int last_B = 0, nsplit = 16;
    std::vector<void *> owned;
};

__global__ void mt_transpose_kernel(const float *__restrict__ w, float *__restrict__ wt, int O, int K)   // (O,K) -> (K,O)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= O * K) return;
    const int k = i / O, o = i - k * O;
    wt[i] = w[(size_t)o * K + k];
}
// mse over all outputs; dy = 2 (y - t) / (B*N) * scale, gated by the ReLU part of the split head
__global__ void mt_loss_kernel(const float *__restrict__ y, const float *__restrict__ t, float *__restrict__ dy, float *__restrict__ lpart,
                               long n, int N, int n_lin, float inv, float scale)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float a = 0.0f;
    if (i < n) {
        const float e = y[i] - t[i];
        a = e * e * inv * scale;
        const int c = (int)(i % N);
        dy[i] = (c >= n_lin && !(y[i] > 0.0f)) ? 0.0f : 2.0f * e * inv * scale;
    }
    __shared__ float sm[256];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) lpart[blockIdx.x] = sm[0];
}
__global__ void mt_sum_kernel(const float *__restrict__ lpart, int n, float *__restrict__ out)
{
    __shared__ float sm[256];
    float a = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) a += lpart[i];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = sm[0];
}
__global__ void mt_adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m1, float *__restrict__ m2,
                               size_t n, float lr_t, float b1, float b2, float eps)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float a = b1 * m1[i] + (1.0f - b1) * gi, v = b2 * m2[i] + (1.0f - b2) * gi * gi;
    m1[i] = a; m2[i] = v;
    p[i] -= lr_t * a / (sqrtf(v) + eps);      // keras: lr_t = lr sqrt(1-b2^t)/(1-b1^t)
}

extern "C" int csa_mlp_train_create(int nlayers, const int *dims, const float *const *weights, const float *const *biases,
                                    float leaky_alpha, int n_lin_out, int max_batch, csa_mlp_trainer **out)
{
    if (nlayers <= 0 || !dims || !weights || !biases || !out || max_batch <= 0) { csa_set_error_msg("csa_mlp_train_create: bad argument"); return CSA_ERR_ARG; }
    for (int l = 0; l <= nlayers; ++l)
        if (dims[l] % 4) { csa_set_error_msg("csa_mlp_train_create: layer widths must be multiples of 4"); return CSA_ERR_UNSUPPORTED; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { csa_set_error_msg("csa_mlp_train_create: no HIP device"); return CSA_ERR_HIP; }
    csa_mlp_trainer *h = new csa_mlp_trainer();
    h->nlayers = nlayers; h->max_batch = max_batch; h->n_lin = n_lin_out; h->alpha = leaky_alpha;
    h->dims.assign(dims, dims + nlayers + 1);
    size_t off = 0, wmax = 0;
    int dmax = 0;
    std::vector<float> host;
    for (int l = 0; l < nlayers; ++l) {
        const size_t nw = (size_t)dims[l + 1] * dims[l];
        h->w_off.push_back(off); host.insert(host.end(), weights[l], weights[l] + nw); off += nw;
        h->b_off.push_back(off); host.insert(host.end(), biases[l], biases[l] + dims[l + 1]); off += dims[l + 1];
        wmax = nw > wmax ? nw : wmax;
        dmax = dims[l + 1] > dmax ? dims[l + 1] : dmax;
    }
    dmax = dims[0] > dmax ? dims[0] : dmax;
    h->n_params = off;
    int rc = CSA_OK;
    auto alloc = [&](size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, sizeof(float) * (n ? n : 1)) != hipSuccess) { rc = CSA_ERR_NOMEM; return (float *)nullptr; }
        h->owned.push_back(p);
        if (hipMemset(p, 0, sizeof(float) * (n ? n : 1)) != hipSuccess) rc = CSA_ERR_HIP;
        return (float *)p;
    };
    h->params = alloc(off); h->m1 = alloc(off); h->m2 = alloc(off);
    if (rc == CSA_OK && hipMemcpy(h->params, host.data(), sizeof(float) * off, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
    h->wT = alloc(wmax);
    for (int l = 0; l < nlayers; ++l) h->act.push_back(alloc((size_t)max_batch * dims[l + 1]));
    h->g0 = alloc((size_t)max_batch * dmax); h->g1 = alloc((size_t)max_batch * dmax);
    h->part = alloc((size_t)h->nsplit * wmax > (size_t)128 * dmax ? (size_t)h->nsplit * wmax : (size_t)128 * dmax);
    h->lpart = alloc(((size_t)max_batch * dims[nlayers] + 255) / 256);
    if (rc) { for (void *p : h->owned) (void)hipFree(p); delete h; csa_set_error_msg("csa_mlp_train_create: allocation failed"); return rc; }
    *out = h;
    return CSA_OK;
}
extern "C" int csa_mlp_train_destroy(csa_mlp_trainer *h)
{
    if (!h) return CSA_ERR_ARG;
    for (void *p : h->owned) (void)hipFree(p);
    delete h;
    return CSA_OK;
}
extern "C" long csa_mlp_train_num_params(const csa_mlp_trainer *h) { return h ? (long)h->n_params : CSA_ERR_ARG; }
extern "C" int csa_mlp_train_copy_params(csa_mlp_trainer *h, int dir, float *buf, void *stream)     // 0 out, 1 in
{
    if (!h || !buf || dir < 0 || dir > 1) return CSA_ERR_ARG;
    CSA_HIP_CHECK(hipMemcpyAsync(dir ? h->params : buf, dir ? buf : h->params, sizeof(float) * h->n_params, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CSA_OK;
}

// x (B, dims[0]) must stay valid until backward; y (B, dims[nlayers]) optional copy of the output
extern "C" int csa_mlp_train_forward(csa_mlp_trainer *h, int B, const float *x, float *y, void *stream)
{
    if (!h || !x || B <= 0 || B > h->max_batch) { csa_set_error_msg("csa_mlp_train_forward: bad argument"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const float *in = x;
    int rc;
    for (int l = 0; l < h->nlayers; ++l) {
        const bool last = l + 1 == h->nlayers;
        if ((rc = launch_gemm_act(in, h->params + h->w_off[l], h->params + h->b_off[l], h->act[l], B, h->dims[l + 1], h->dims[l],
                                  last ? 2 : 1, h->alpha, last ? h->n_lin : 0, s))) return rc;
        in = h->act[l];
    }
    if (y) CSA_HIP_CHECK(hipMemcpyAsync(y, in, sizeof(float) * (size_t)B * h->dims[h->nlayers], hipMemcpyDeviceToDevice, s));
    h->x_in = x; h->last_B = B;
    return CSA_OK;
}

extern "C" int csa_mlp_train_backward(csa_mlp_trainer *h, const float *y_true, float grad_scale, float *loss_out, float *grads, void *stream)
{
    if (!h || !y_true || !grads || h->last_B <= 0) { csa_set_error_msg("csa_mlp_train_backward: bad argument / no forward"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int B = h->last_B, NL = h->nlayers, N = h->dims[NL];
    const long n = (long)B * N;
    int rc;
    CSA_HIP_CHECK(hipMemsetAsync(grads, 0, sizeof(float) * h->n_params, s));
    const int nb = (int)((n + 255) / 256);
    float *d = h->g0, *dn = h->g1;
    hipLaunchKernelGGL(mt_loss_kernel, dim3(nb), dim3(256), 0, s, h->act[NL - 1], y_true, d, h->lpart, n, N, h->n_lin, 1.0f / (float)n, grad_scale);
    if (loss_out) hipLaunchKernelGGL(mt_sum_kernel, dim3(1), dim3(256), 0, s, h->lpart, nb, loss_out);
    for (int l = NL - 1; l >= 0; --l) {
        const int O = h->dims[l + 1], K = h->dims[l];
        const float *a_in = l == 0 ? h->x_in : h->act[l - 1];
        // dW (O,K) = d^T a_in, db = column sums of d
        if ((rc = launch_gemm_tn_partial(d, O, a_in, K, h->part, B, O, K, h->nsplit, s))) return rc;
        if ((rc = launch_reduce_partials(h->part, h->nsplit, O * K, nullptr, nullptr, grads + h->w_off[l], s))) return rc;
        const int cs = B >= 128 ? 128 : 1;
        if ((rc = launch_colsum_partial(d, h->part, B, O, cs, s))) return rc;
        if ((rc = launch_reduce_partials(h->part, cs, O, nullptr, nullptr, grads + h->b_off[l], s))) return rc;
        if (l == 0) break;
        // d_prev (B,K) = (d W) * LeakyReLU'(pre_{l-1}), the derivative taken from the saved output's sign
        hipLaunchKernelGGL(mt_transpose_kernel, dim3((O * K + 255) / 256), dim3(256), 0, s, h->params + h->w_off[l], h->wT, O, K);
        GemmEpi e{};
        e.gate = h->act[l - 1]; e.gscale = 1.0f; e.gneg = h->alpha;
        if ((rc = launch_gemm_epi(d, h->wT, nullptr, dn, B, K, O, 0, 0.0f, 0, O, K, 0, 0, e, s))) return rc;
        float *t = d; d = dn; dn = t;
    }
    return CSA_OK;
}

extern "C" int csa_mlp_train_adam(csa_mlp_trainer *h, const float *grads, float lr, float beta1, float beta2, float eps, int step, void *stream)
{
    if (!h || !grads || step <= 0) { csa_set_error_msg("csa_mlp_train_adam: bad argument"); return CSA_ERR_ARG; }
    const float lr_t = lr * sqrtf(1.0f - powf(beta2, (float)step)) / (1.0f - powf(beta1, (float)step));
    hipLaunchKernelGGL(mt_adam_kernel, dim3((unsigned)((h->n_params + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->params, grads, h->m1,
                       h->m2, h->n_params, lr_t, beta1, beta2, eps);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
