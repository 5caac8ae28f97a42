// cnn_api.hip -- the offline Keras CNN baseline, forward:
//   baseline_models/CNN/training/hpo_train.py:124-200
//   in (60,6) -> 12 x [Conv1D(406,3,same)+ReLU, Conv1D(406,3,same)+ReLU, + Conv1D(406,1)(block input)]
//             -> Conv1D(10,1,elu) -> Dense(2,linear) || Dense(8,relu) -> (60,10);  ~1.58 GFLOP per column.
// Every convolution is an implicit GEMM on the fp32 matrix cores (gemm.hip, conv mode): activations are kept
// channels-last as (column*level, C) rows with C padded to a multiple of 8 (406 -> 408), so the three taps of
// a row are ONE contiguous 3*C run of the previous layer's output and no im2col buffer exists; bias, ReLU / ELU
// and the residual add are fused into the GEMM epilogue.  Dropout is inference-mode identity.
#include "common.h"
#include <vector>

struct CnnLayer { float *w, *b; int cin_p, cout_p, k; };

struct csa_cnn {
    int depth, L, cin, width, cout, n_lin, max_batch, cin_p, wp, cout_p;
    std::vector<CnnLayer> conv_a, conv_b, conv_r;
    CnnLayer pre_out, dense;
    float *buf[3], *xin, *y10;
    std::vector<void *> owned;
};

static int rup(int v, int m) { return (v + m - 1) / m * m; }

__global__ void pad_channels_kernel(const float *__restrict__ x, float *__restrict__ y, int rows, int c, int cp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cp) return;
    const int r = i / cp, ch = i - r * cp;
    y[i] = ch < c ? x[(size_t)r * c + ch] : 0.0f;
}

// weights: PyTorch Conv1d layout (cout, cin, k), HOST pointers, in the order
//   for each block: conv_a (width,cin_i,3), conv_b (width,width,3), conv_res (width,cin_i,1)
//   then pre_out (cout, width, 1), then dense (cout, cout) = [Dense(n_lin) ; Dense(cout-n_lin)] stacked
extern "C" int csa_cnn_create(int depth, int nlev, int cin, int width, int cout, int n_lin,
                              const float *const *weights, const float *const *biases, int max_batch, csa_cnn **out)
{
    if (depth <= 0 || !weights || !biases || !out || max_batch <= 0) { csa_set_error_msg("csa_cnn_create: bad argument"); return CSA_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { csa_set_error_msg("csa_cnn_create: no HIP device"); return CSA_ERR_HIP; }
    csa_cnn *h = new csa_cnn();
    h->depth = depth; h->L = nlev; h->cin = cin; h->width = width; h->cout = cout; h->n_lin = n_lin; h->max_batch = max_batch;
    h->cin_p = rup(cin, 8); h->wp = rup(width, 8); h->cout_p = rup(cout, 4);
    int rc = CSA_OK;
    auto up = [&](const float *src, size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, sizeof(float) * (n ? n : 1)) != hipSuccess) { rc = CSA_ERR_NOMEM; return (float *)nullptr; }
        h->owned.push_back(p);
        if (src && hipMemcpy(p, src, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
        return (float *)p;
    };
    // (co, ci, k) -> GEMM weight (co_p rows, k*ci_p): W[n][d*ci_p + c] = w[n][c][d]; padded rows / columns zero
    auto pack = [&](const float *w, const float *b, int co, int ci, int k, int co_p, int ci_p) {
        std::vector<float> g((size_t)co_p * k * ci_p, 0.0f), bb(co_p, 0.0f);
        for (int n = 0; n < co; ++n) {
            for (int c = 0; c < ci; ++c)
                for (int d = 0; d < k; ++d) g[(size_t)n * k * ci_p + (size_t)d * ci_p + c] = w[((size_t)n * ci + c) * k + d];
            bb[n] = b[n];
        }
        CnnLayer l;
        l.w = up(g.data(), g.size()); l.b = up(bb.data(), bb.size()); l.cin_p = ci_p; l.cout_p = co_p; l.k = k;
        return l;
    };
    int wi = 0;
    for (int i = 0; i < depth; ++i) {
        const int ci = i == 0 ? cin : width, ci_p = i == 0 ? h->cin_p : h->wp;
        h->conv_a.push_back(pack(weights[wi], biases[wi], width, ci, 3, h->wp, ci_p)); ++wi;
        h->conv_b.push_back(pack(weights[wi], biases[wi], width, width, 3, h->wp, h->wp)); ++wi;
        h->conv_r.push_back(pack(weights[wi], biases[wi], width, ci, 1, h->wp, ci_p)); ++wi;
    }
    h->pre_out = pack(weights[wi], biases[wi], cout, width, 1, h->cout_p, h->wp); ++wi;
    h->dense = pack(weights[wi], biases[wi], cout, cout, 1, cout, h->cout_p); ++wi;
    const size_t rows = (size_t)max_batch * nlev;
    for (int i = 0; i < 3; ++i) h->buf[i] = up(nullptr, rows * h->wp);
    h->xin = up(nullptr, rows * h->cin_p);
    h->y10 = up(nullptr, rows * h->cout_p);
    if (rc) { for (void *p : h->owned) (void)hipFree(p); delete h; return rc; }
    *out = h;
    return CSA_OK;
}

extern "C" int csa_cnn_destroy(csa_cnn *h)
{
    if (!h) return CSA_ERR_ARG;
    for (void *p : h->owned) (void)hipFree(p);
    delete h;
    return CSA_OK;
}

// x (B, nlev, cin) -> y (B, nlev, cout)
extern "C" int csa_cnn_forward(csa_cnn *h, int B, const float *x, float *y, void *stream)
{
    if (!h || !x || !y || B <= 0 || B > h->max_batch) { csa_set_error_msg("csa_cnn_forward: bad argument"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int M = B * h->L, L = h->L, wp = h->wp;
    hipLaunchKernelGGL(pad_channels_kernel, dim3((M * h->cin_p + 255) / 256), dim3(256), 0, s, x, h->xin, M, h->cin, h->cin_p);
    const float *in = h->xin;
    int in_c = h->cin_p, cur = -1, rc;
    for (int i = 0; i < h->depth; ++i) {
        // two scratch buffers different from the one holding the block input (cur = -1: the padded input)
        const int a = (cur + 1) % 3, b2 = (cur + 2) % 3;
        float *t1 = h->buf[a], *t2 = h->buf[b2];
        const CnnLayer &ca = h->conv_a[i], &cb = h->conv_b[i], &cr = h->conv_r[i];
        if ((rc = launch_gemm_ex(in, ca.w, ca.b, t1, M, wp, 3 * in_c, /*relu*/ 1, 0.0f, 0, in_c, wp, L, in_c, 0, s))) return rc;
        if ((rc = launch_gemm_ex(t1, cb.w, cb.b, t2, M, wp, 3 * wp, 1, 0.0f, 0, wp, wp, L, wp, 0, s))) return rc;
        if ((rc = launch_gemm_ex(in, cr.w, cr.b, t2, M, wp, in_c, 0, 0.0f, 0, in_c, wp, 0, 0, /*accumulate*/ 1, s))) return rc;
        in = t2; in_c = wp; cur = b2;
    }
    if ((rc = launch_gemm_ex(in, h->pre_out.w, h->pre_out.b, h->y10, M, h->cout_p, wp, /*elu*/ 3, 0.0f, 0, wp, h->cout_p, 0, 0, 0, s))) return rc;
    return launch_gemm_ex(h->y10, h->dense.w, h->dense.b, y, M, h->cout, h->cout_p, /*split*/ 2, 0.0f, h->n_lin, h->cout_p,
                          h->cout, 0, 0, 0, s);
}

// ---- data-format adapters either side of the CNN (climsim_utils/data_utils.py:2104-2175, V1 variables) ----------------
// flat (N, nprof*nlev + nscal) -> channels-last (N, nlev, nprof + nscal): profiles become channels, scalars are repeated
// over the levels (reshape_input_for_cnn: nprof 2, nscal 4; reshape_target_for_cnn: nprof 2, nscal 8)
__global__ void cnn_reshape_to_kernel(const float *__restrict__ x, float *__restrict__ y, long total, int nlev, int nprof, int nscal)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int C = nprof + nscal, W = nprof * nlev + nscal;
    const int c = (int)(i % C);
    const long r = i / C;
    const int l = (int)(r % nlev);
    const long n = r / nlev;
    y[i] = c < nprof ? x[n * W + (long)c * nlev + l] : x[n * W + (long)nprof * nlev + (c - nprof)];
}
// channels-last (N, nlev, nprof + nscal) -> flat (N, nprof*nlev + nscal): scalars = mean over the levels
// (reshape_target_from_cnn)
__global__ void cnn_reshape_from_kernel(const float *__restrict__ y, float *__restrict__ x, long total, int nlev, int nprof, int nscal)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int C = nprof + nscal, W = nprof * nlev + nscal;
    const int w = (int)(i % W);
    const long n = i / W;
    if (w < nprof * nlev) {
        x[i] = y[(n * nlev + w % nlev) * C + w / nlev];
    } else {
        const int c = nprof + (w - nprof * nlev);
        float a = 0.0f;
        for (int l = 0; l < nlev; ++l) a += y[(n * nlev + l) * C + c];
        x[i] = a / (float)nlev;
    }
}
extern "C" int csa_cnn_reshape_to(int N, int nlev, int nprof, int nscal, const float *flat, float *chan, void *stream)
{
    if (N <= 0 || nlev <= 0 || nprof < 0 || nscal < 0 || !flat || !chan) { csa_set_error_msg("csa_cnn_reshape_to: bad argument"); return CSA_ERR_ARG; }
    const long total = (long)N * nlev * (nprof + nscal);
    hipLaunchKernelGGL(cnn_reshape_to_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat, chan, total, nlev, nprof, nscal);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
extern "C" int csa_cnn_reshape_from(int N, int nlev, int nprof, int nscal, const float *chan, float *flat, void *stream)
{
    if (N <= 0 || nlev <= 0 || nprof < 0 || nscal < 0 || !flat || !chan) { csa_set_error_msg("csa_cnn_reshape_from: bad argument"); return CSA_ERR_ARG; }
    const long total = (long)N * (nprof * nlev + nscal);
    hipLaunchKernelGGL(cnn_reshape_from_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, chan, flat, total, nlev, nprof, nscal);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
