// online.hip -- the "generic online" boundary variant of SURVEY.md section 8(b): forward(x (B, n_in)) -> (B, 368),
// the wrapper the host climate model loads for the MLP_v2rh / v4 emulators:
//   online_testing/model_postprocessing/v4_nn_wrapper.ipynb cell 5 (class NewModel: preprocessing / forward /
//     postprocessing on the flat input vector), online_testing/README.md:47-50 (the host contract);
//   online_testing/baseline_models/MLP_v2rh/training/mlp.py:25-67 (the wrapped MLP: Linear+ReLU hidden layers, final
//     Linear, output_prune of the top strato_lev_out levels of four tendency profiles, ReLU on the last 8 outputs).
// Three pieces on one stream:
//   online_prep_kernel  (HBM-bound, one pass over the input): per input COLUMN j a table entry decides
//        x = lbd[j] != 0 ? 1 - exp(-x * lbd[j]) : x            (cloud liquid / ice exponential transform)
//        x = (x - sub[j]) / div[j];  NaN, +-Inf -> 0             (normalisation and scrub, in the reference's order)
//        flag bit 0: x = 0   (pruned stratospheric levels)       flag bit 1: x = clamp(x, lo, hi)   (RH clip)
//      written into a buffer whose row stride is n_in rounded up to 4 (zero padding; the first layer's weight rows are
//      padded the same way at create time, so the GEMM's float4 operand loads stay aligned for n_in = 557 or 1525);
//   the GEMM chain of mlp_api.hip (ReLU = LeakyReLU with slope 0 in the epilogue, split linear | ReLU head);
//   online_post_kernel  (HBM-bound): y = (keep[j] ? y : 0) / out_scale[j]   (mlp.py output_prune + NewModel.postprocessing).
// The caller's x is never written (the reference wrapper normalises its argument in place; a drop-in must not rely on it).
#include "common.h"
#include <cmath>
#include <vector>

struct csa_mlp;
extern "C" int csa_mlp_create(int nlayers, const int *dims, const float *const *weights, const float *const *biases,
                              float leaky_alpha, int n_lin_out, int max_batch, csa_mlp **out);
extern "C" int csa_mlp_destroy(csa_mlp *h);
extern "C" int csa_mlp_forward(csa_mlp *h, int B, const float *x, float *y, void *stream);

struct csa_online {
    int n_in, n_in_pad, n_out, max_batch;
    float clip_lo, clip_hi;
    float *sub, *div, *lbd, *oscale;
    unsigned char *iflags, *okeep;
    float *xp;                    // (max_batch, n_in_pad)
    csa_mlp *mlp;
    std::vector<void *> owned;
};

__global__ __launch_bounds__(256) void online_prep_kernel(const float *__restrict__ x, float *__restrict__ xp,
                                                          const float *__restrict__ sub, const float *__restrict__ div,
                                                          const float *__restrict__ lbd, const unsigned char *__restrict__ fl,
                                                          int B, int n_in, int n_pad, float lo, float hi)
{
    // one thread per PADDED element: consecutive lanes write consecutive addresses of xp and read (almost) consecutive ones of x
    const long total = (long)B * n_pad;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int b = (int)(e / n_pad), j = (int)(e - (long)b * n_pad);
        float v = 0.0f;
        if (j < n_in) {
            v = x[(size_t)b * n_in + j];
            const float l = lbd[j];
            if (l != 0.0f) v = 1.0f - expf(-v * l);
            v = (v - sub[j]) / div[j];
            if (!(fabsf(v) <= 3.402823466e38f)) v = 0.0f;      // NaN and +-Inf (torch.where(isnan) ; torch.where(isinf))
            const unsigned f = fl[j];
            if (f & 1u) v = 0.0f;
            if (f & 2u) v = fminf(fmaxf(v, lo), hi);
        }
        xp[e] = v;
    }
}

__global__ __launch_bounds__(256) void online_post_kernel(float *__restrict__ y, const float *__restrict__ oscale,
                                                          const unsigned char *__restrict__ keep, int B, int n_out)
{
    const long total = (long)B * n_out;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e % n_out);
        y[e] = (keep[j] ? y[e] : 0.0f) / oscale[j];
    }
}

extern "C" int csa_online_create(int n_in, int nlayers, const int *dims, const float *const *weights,
                                 const float *const *biases, const float *in_sub, const float *in_div,
                                 const float *in_lbd, const unsigned char *in_flags, float clip_lo, float clip_hi,
                                 const float *out_scale, const unsigned char *out_keep, int n_relu_tail, int max_batch,
                                 csa_online **out)
{
    if (n_in <= 0 || nlayers <= 0 || !dims || !weights || !biases || !in_sub || !in_div || !in_lbd || !in_flags ||
        !out_scale || !out_keep || !out || max_batch <= 0 || dims[0] != n_in || n_relu_tail < 0 || n_relu_tail > dims[nlayers]) {
        csa_set_error_msg("csa_online_create: bad argument");
        return CSA_ERR_ARG;
    }
    const int n_pad = (n_in + 3) & ~3, n_out = dims[nlayers];
    // first layer: pad the rows of W0 (dims[1], n_in) to the padded input stride
    std::vector<float> w0((size_t)dims[1] * n_pad, 0.0f);
    for (int r = 0; r < dims[1]; ++r)
        for (int k = 0; k < n_in; ++k) w0[(size_t)r * n_pad + k] = weights[0][(size_t)r * n_in + k];
    std::vector<int> d(dims, dims + nlayers + 1);
    d[0] = n_pad;
    std::vector<const float *> wp(weights, weights + nlayers);
    wp[0] = w0.data();
    csa_online *h = new csa_online();
    h->n_in = n_in; h->n_in_pad = n_pad; h->n_out = n_out; h->max_batch = max_batch;
    h->clip_lo = clip_lo; h->clip_hi = clip_hi; h->mlp = nullptr;
    int rc = csa_mlp_create(nlayers, d.data(), wp.data(), biases, 0.0f, n_out - n_relu_tail, max_batch, &h->mlp);
    if (rc != CSA_OK) { delete h; return rc; }
    auto up = [&](const void *src, size_t bytes) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { rc = CSA_ERR_NOMEM; return (void *)nullptr; }
        h->owned.push_back(p);
        if (src && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
        return p;
    };
    h->sub = (float *)up(in_sub, sizeof(float) * n_in);
    h->div = (float *)up(in_div, sizeof(float) * n_in);
    h->lbd = (float *)up(in_lbd, sizeof(float) * n_in);
    h->iflags = (unsigned char *)up(in_flags, n_in);
    h->oscale = (float *)up(out_scale, sizeof(float) * n_out);
    h->okeep = (unsigned char *)up(out_keep, n_out);
    h->xp = (float *)up(nullptr, sizeof(float) * (size_t)max_batch * n_pad);
    if (rc != CSA_OK) {
        for (void *p : h->owned) (void)hipFree(p);
        csa_mlp_destroy(h->mlp);
        delete h;
        return rc;
    }
    *out = h;
    return CSA_OK;
}

extern "C" int csa_online_destroy(csa_online *h)
{
    if (!h) return CSA_ERR_ARG;
    for (void *p : h->owned) (void)hipFree(p);
    if (h->mlp) csa_mlp_destroy(h->mlp);
    delete h;
    return CSA_OK;
}

extern "C" int csa_online_dims(const csa_online *h, int *n_in, int *n_out)
{
    if (!h) return CSA_ERR_ARG;
    if (n_in) *n_in = h->n_in;
    if (n_out) *n_out = h->n_out;
    return CSA_OK;
}

// x (B, n_in) raw physical units -> y (B, n_out) raw tendencies; device pointers; x is not modified
extern "C" int csa_online_forward(csa_online *h, int B, const float *x, float *y, void *stream)
{
    if (!h || !x || !y || B <= 0 || B > h->max_batch) { csa_set_error_msg("csa_online_forward: bad argument"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const long tin = (long)B * h->n_in_pad, tout = (long)B * h->n_out;
    const int gin = (int)((tin + 255) / 256 < 4096 ? (tin + 255) / 256 : 4096);
    const int gout = (int)((tout + 255) / 256 < 4096 ? (tout + 255) / 256 : 4096);
    hipLaunchKernelGGL(online_prep_kernel, dim3(gin), dim3(256), 0, s, x, h->xp, h->sub, h->div, h->lbd, h->iflags, B,
                       h->n_in, h->n_in_pad, h->clip_lo, h->clip_hi);
    CSA_HIP_CHECK(hipGetLastError());
    int rc = csa_mlp_forward(h->mlp, B, h->xp, y, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(online_post_kernel, dim3(gout), dim3(256), 0, s, y, h->oscale, h->okeep, B, h->n_out);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
