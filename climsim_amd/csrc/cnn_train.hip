// cnn_train.hip -- one optimiser step of the Keras CNN baseline (BASELINE.json configs[3]):
//   baseline_models/CNN/training/hpo_train.py:124-236  ResNet-1D, Dropout(0.175) after both activations of a block,
//   loss mae_adjusted (:118-120), keras.optimizers.Adam.
// Every convolution, its input gradient (a convolution of dY with the flipped, transposed kernel) and its weight
// gradient (dY^T * im2col(X)) is an implicit GEMM on the fp32 matrix cores; im2col never exists in memory (rows of a
// channels-last activation are contiguous, so the three taps of a level are one 3*C run).  Bias, ReLU, dropout mask,
// residual add and the ReLU/dropout backward gate are GEMM epilogues.  Parameters live in ONE flat device buffer in
// the GEMM layout (cout_p rows x k*cin_p, channels padded to multiples of 8; padding stays exactly zero under
// Adam because its gradient is exactly zero), gradients in a matching flat buffer -> one RCCL all-reduce per step.
#include "common.h"
#include "train.h"
#include <vector>

struct CtLayer { int cout, cin, k, cout_p, cin_p; size_t w_off, b_off; };

struct csa_cnn_trainer {
    int depth, L, cin, width, cout, n_lin, max_batch, cin_p, wp, cout_p;
    float dropout;
    std::vector<CtLayer> layers;      // a0, b0, r0, a1, ..., pre_out, dense
    size_t n_params = 0;
    float *params = nullptr, *m1 = nullptr, *m2 = nullptr;
    float *wflip = nullptr;           // transposed + tap-flipped weights of one layer (input-gradient GEMM)
    float *xin = nullptr, *z = nullptr, *y = nullptr, *dy = nullptr, *dzp = nullptr, *ddp = nullptr;
    std::vector<float *> t1, t2, hb;  // saved activations per block
    float *g0 = nullptr, *g1 = nullptr, *g2 = nullptr;   // gradient ping-pong buffers (M x wp)
    float *part = nullptr;            // split-M partials of the weight-gradient GEMMs
    float *lpart = nullptr;           // loss partial sums
    size_t part_floats = 0;
    int nsplit = 16;
    int last_B = 0;
    std::vector<void *> owned;
};

static int ct_rup(int v, int m) { return (v + m - 1) / m * m; }

__global__ void ct_pad_kernel(const float *__restrict__ x, float *__restrict__ y, int rows, int c, int cp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cp) return;
    const int r = i / cp, ch = i - r * cp;
    y[i] = ch < c ? x[(size_t)r * c + ch] : 0.0f;
}

// W (cout_p, k*cin_p) -> W' (cin_p, k*cout_p):  W'[c][d*cout_p + n] = W[n][(k-1-d)*cin_p + c]
__global__ void ct_flip_kernel(const float *__restrict__ w, float *__restrict__ wf, int cout_p, int cin_p, int k)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cout_p * cin_p * k) return;
    const int c = i / (k * cout_p), rem = i - c * (k * cout_p), d = rem / cout_p, n = rem - d * cout_p;
    wf[i] = w[(size_t)n * (k * cin_p) + (size_t)(k - 1 - d) * cin_p + c];
}

// g = dh * (t > 0 ? scale : 0)    (backward of ReLU followed by inverted dropout, from the saved output t)
__global__ void ct_gate_kernel(const float *__restrict__ dh, const float *__restrict__ t, float *__restrict__ g, size_t n, float scale)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] = t[i] > 0.0f ? dh[i] * scale : 0.0f;
}

// mae_adjusted (hpo_train.py:118-120): mean|e|[:,:,0:n_lin]*(120/128) + mean|e|[:,:,n_lin:]*(8/128); per-row partials + dy
__global__ void ct_loss_kernel(const float *__restrict__ y, const float *__restrict__ yt, float *__restrict__ dy,
                               float *__restrict__ lpart, int M, int cout, int n_lin, float w_lin, float w_rest)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float a = 0.0f;
    if (r < M) {
        for (int c = 0; c < cout; ++c) {
            const float e = y[(size_t)r * cout + c] - yt[(size_t)r * cout + c];
            const float wgt = c < n_lin ? w_lin : w_rest;
            a += fabsf(e) * wgt;
            dy[(size_t)r * cout + c] = e > 0.0f ? wgt : (e < 0.0f ? -wgt : 0.0f);
        }
    }
    __shared__ float sm[256];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) lpart[blockIdx.x] = sm[0];
}
__global__ void ct_loss_final_kernel(const float *__restrict__ lpart, int n, float *__restrict__ out)
{
    __shared__ float sm[256];
    float a = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) a += lpart[i];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = sm[0];
}

// backward of the two tiny output layers, one thread per (column, level) row:
//   y = [lin | relu](Wd z + bd), z = elu(pre):   ddp = dy gated by the split head (padded to cp), dzp = (Wd^T ddp) * elu'(pre)
__global__ void ct_tail_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, const float *__restrict__ z,
                                   const float *__restrict__ wd, float *__restrict__ ddp, float *__restrict__ dzp,
                                   int M, int cout, int cp, int n_lin)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    float d[16];
    for (int c = 0; c < cp; ++c) {
        float v = 0.0f;
        if (c < cout) {
            v = dy[(size_t)r * cout + c];
            if (c >= n_lin && !(y[(size_t)r * cout + c] > 0.0f)) v = 0.0f;
        }
        d[c] = v;
        ddp[(size_t)r * cp + c] = v;
    }
    for (int j = 0; j < cp; ++j) {
        float a = 0.0f;
        for (int c = 0; c < cout; ++c) a += d[c] * wd[(size_t)c * cp + j];
        const float zj = z[(size_t)r * cp + j];
        dzp[(size_t)r * cp + j] = j < cout ? a * (zj > 0.0f ? 1.0f : zj + 1.0f) : 0.0f;
    }
}

// keras.optimizers.Adam: p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
__global__ void ct_adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m1, float *__restrict__ m2,
                               size_t n, float lr_t, float b1, float b2, float eps)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float a = b1 * m1[i] + (1.0f - b1) * gi, v = b2 * m2[i] + (1.0f - b2) * gi * gi;
    m1[i] = a; m2[i] = v;
    p[i] -= lr_t * a / (sqrtf(v) + eps);
}

extern "C" int csa_cnn_train_create(int depth, int nlev, int cin, int width, int cout, int n_lin, const float *const *weights,
                                    const float *const *biases, int max_batch, float dropout, csa_cnn_trainer **out)
{
    if (depth <= 0 || !weights || !biases || !out || max_batch <= 0 || cout > 16 || dropout < 0.0f || dropout >= 1.0f) {
        csa_set_error_msg("csa_cnn_train_create: bad argument");
        return CSA_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { csa_set_error_msg("csa_cnn_train_create: no HIP device"); return CSA_ERR_HIP; }
    csa_cnn_trainer *h = new csa_cnn_trainer();
    h->depth = depth; h->L = nlev; h->cin = cin; h->width = width; h->cout = cout; h->n_lin = n_lin; h->max_batch = max_batch;
    h->cin_p = ct_rup(cin, 8); h->wp = ct_rup(width, 8); h->cout_p = ct_rup(cout, 4); h->dropout = dropout;
    size_t off = 0;
    auto add = [&](int co, int ci, int k, int co_p, int ci_p) {
        CtLayer l{co, ci, k, co_p, ci_p, off, 0};
        off += (size_t)co_p * k * ci_p;
        l.b_off = off;
        off += co_p;
        h->layers.push_back(l);
    };
    for (int i = 0; i < depth; ++i) {
        const int ci = i == 0 ? cin : width, ci_p = i == 0 ? h->cin_p : h->wp;
        add(width, ci, 3, h->wp, ci_p); add(width, width, 3, h->wp, h->wp); add(width, ci, 1, h->wp, ci_p);
    }
    add(cout, width, 1, h->cout_p, h->wp);
    add(cout, cout, 1, h->cout_p, h->cout_p);
    h->n_params = off;
    std::vector<float> host(off, 0.0f);
    for (size_t li = 0; li < h->layers.size(); ++li) {
        const CtLayer &l = h->layers[li];
        for (int n = 0; n < l.cout; ++n) {
            for (int c = 0; c < l.cin; ++c)
                for (int d = 0; d < l.k; ++d)
                    host[l.w_off + (size_t)n * l.k * l.cin_p + (size_t)d * l.cin_p + c] = weights[li][((size_t)n * l.cin + c) * l.k + d];
            host[l.b_off + n] = biases[li][n];
        }
    }
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
    const size_t rows = (size_t)max_batch * nlev, act = rows * h->wp;
    h->wflip = alloc((size_t)h->wp * 3 * h->wp);
    h->xin = alloc(rows * h->cin_p);
    for (int i = 0; i < depth; ++i) { h->t1.push_back(alloc(act)); h->t2.push_back(alloc(act)); h->hb.push_back(alloc(act)); }
    h->z = alloc(rows * h->cout_p); h->y = alloc(rows * cout); h->dy = alloc(rows * cout);
    h->dzp = alloc(rows * h->cout_p); h->ddp = alloc(rows * h->cout_p);
    h->g0 = alloc(act); h->g1 = alloc(act); h->g2 = alloc(act);
    h->part_floats = (size_t)h->nsplit * h->wp * 3 * h->wp;
    h->part = alloc(h->part_floats);
    h->lpart = alloc((rows + 255) / 256);
    if (rc) { for (void *p : h->owned) (void)hipFree(p); delete h; csa_set_error_msg("csa_cnn_train_create: allocation failed"); return rc; }
    *out = h;
    return CSA_OK;
}

extern "C" int csa_cnn_train_destroy(csa_cnn_trainer *h)
{
    if (!h) return CSA_ERR_ARG;
    for (void *p : h->owned) (void)hipFree(p);
    delete h;
    return CSA_OK;
}
extern "C" long csa_cnn_train_num_params(const csa_cnn_trainer *h) { return h ? (long)h->n_params : CSA_ERR_ARG; }
extern "C" int csa_cnn_train_num_layers(const csa_cnn_trainer *h) { return h ? (int)h->layers.size() : CSA_ERR_ARG; }
extern "C" float *csa_cnn_train_params(csa_cnn_trainer *h) { return h ? h->params : nullptr; }
// copy the flat parameters to / from a caller-owned device buffer (export, checkpoint restore, broadcast)
extern "C" int csa_cnn_train_get_params(csa_cnn_trainer *h, float *dst, void *stream)
{
    if (!h || !dst) return CSA_ERR_ARG;
    CSA_HIP_CHECK(hipMemcpyAsync(dst, h->params, sizeof(float) * h->n_params, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CSA_OK;
}
extern "C" int csa_cnn_train_set_params(csa_cnn_trainer *h, const float *src, void *stream)
{
    if (!h || !src) return CSA_ERR_ARG;
    CSA_HIP_CHECK(hipMemcpyAsync(h->params, src, sizeof(float) * h->n_params, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CSA_OK;
}
// debug tap: saved activation of the last forward, which = 0 (t1: after conv_a+ReLU+dropout), 1 (t2), 2 (block output);
// dst (B*nlev, width_padded) device buffer; *ld receives the padded width
extern "C" int csa_cnn_train_get_act(csa_cnn_trainer *h, int block, int which, float *dst, int *ld, void *stream)
{
    if (!h || !dst || block < 0 || block >= h->depth || which < 0 || which > 2 || h->last_B <= 0) return CSA_ERR_ARG;
    const float *src = which == 0 ? h->t1[block] : (which == 1 ? h->t2[block] : h->hb[block]);
    CSA_HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(float) * (size_t)h->last_B * h->L * h->wp, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (ld) *ld = h->wp;
    return CSA_OK;
}
extern "C" int csa_cnn_train_layer_info(const csa_cnn_trainer *h, int i, long *w_off, long *b_off, int *cout, int *cin, int *k,
                                        int *cout_p, int *cin_p)
{
    if (!h || i < 0 || i >= (int)h->layers.size()) return CSA_ERR_ARG;
    const CtLayer &l = h->layers[i];
    *w_off = (long)l.w_off; *b_off = (long)l.b_off; *cout = l.cout; *cin = l.cin; *k = l.k; *cout_p = l.cout_p; *cin_p = l.cin_p;
    return CSA_OK;
}

// forward in training mode: masks (2*depth, B*nlev, width) uint8 keep-flags in layer order (null = no dropout); y (B,nlev,cout)
extern "C" int csa_cnn_train_forward(csa_cnn_trainer *h, int B, const float *x, const unsigned char *masks, float *y_out, void *stream)
{
    if (!h || !x || B <= 0 || B > h->max_batch) { csa_set_error_msg("csa_cnn_train_forward: bad argument"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int M = B * h->L, L = h->L, wp = h->wp;
    const float ms = 1.0f / (1.0f - h->dropout);
    hipLaunchKernelGGL(ct_pad_kernel, dim3((M * h->cin_p + 255) / 256), dim3(256), 0, s, x, h->xin, M, h->cin, h->cin_p);
    const float *in = h->xin;
    int in_c = h->cin_p, rc;
    for (int i = 0; i < h->depth; ++i) {
        const CtLayer &a = h->layers[3 * i], &b = h->layers[3 * i + 1], &r = h->layers[3 * i + 2];
        GemmEpi e1{}, e2{}, e3{};
        if (masks) {
            e1.mask = masks + (size_t)(2 * i) * M * h->width; e1.ldmask = h->width; e1.mscale = ms;
            e2.mask = masks + (size_t)(2 * i + 1) * M * h->width; e2.ldmask = h->width; e2.mscale = ms;
        }
        if ((rc = launch_gemm_epi(in, h->params + a.w_off, h->params + a.b_off, h->t1[i], M, wp, 3 * in_c, 1, 0.0f, 0, in_c, wp, L, in_c, e1, s))) return rc;
        if ((rc = launch_gemm_epi(h->t1[i], h->params + b.w_off, h->params + b.b_off, h->t2[i], M, wp, 3 * wp, 1, 0.0f, 0, wp, wp, L, wp, e2, s))) return rc;
        e3.addsrc = h->t2[i];
        if ((rc = launch_gemm_epi(in, h->params + r.w_off, h->params + r.b_off, h->hb[i], M, wp, in_c, 0, 0.0f, 0, in_c, wp, 0, 0, e3, s))) return rc;
        in = h->hb[i]; in_c = wp;
    }
    const CtLayer &po = h->layers[3 * h->depth], &de = h->layers[3 * h->depth + 1];
    if ((rc = launch_gemm_ex(in, h->params + po.w_off, h->params + po.b_off, h->z, M, h->cout_p, wp, 3, 0.0f, 0, wp, h->cout_p, 0, 0, 0, s))) return rc;
    if ((rc = launch_gemm_ex(h->z, h->params + de.w_off, h->params + de.b_off, h->y, M, h->cout, h->cout_p, 2, 0.0f, h->n_lin, h->cout_p, h->cout, 0, 0, 0, s))) return rc;
    if (y_out) CSA_HIP_CHECK(hipMemcpyAsync(y_out, h->y, sizeof(float) * M * h->cout, hipMemcpyDeviceToDevice, s));
    h->last_B = B;
    return CSA_OK;
}

static int ct_wgrad(csa_cnn_trainer *h, const CtLayer &l, const float *dY, const float *X, int M, int conv, float *grads, hipStream_t s)
{
    // dW (cout_p, k*cin_p) = dY^T * [im2col] X  and  db = column sums of dY, split over M and reduced deterministically
    int rc;
    const int N1 = l.cout_p, N2 = l.k * l.cin_p;
    if ((rc = launch_gemm_tn_conv(dY, l.cout_p, X, l.cin_p, h->part, M, N1, N2, h->nsplit, conv ? h->L : 0, conv ? l.cin_p : 0, s))) return rc;
    if ((rc = launch_reduce_partials(h->part, h->nsplit, N1 * N2, nullptr, nullptr, grads + l.w_off, s))) return rc;
    const int cs = 128;   // bias gradient: more, shorter row slices than the GEMM split (the partial buffer is far larger)
    if ((rc = launch_colsum_partial(dY, h->part, M, N1, cs, s))) return rc;
    return launch_reduce_partials(h->part, cs, N1, nullptr, nullptr, grads + l.b_off, s);
}

static int ct_dgrad(csa_cnn_trainer *h, const CtLayer &l, const float *dY, float *dX, int M, const GemmEpi &e, hipStream_t s)
{
    // dX (M, cin_p) = conv(dY, flipped kernel): NT GEMM against W' (cin_p, k*cout_p)
    const int n = l.cout_p * l.cin_p * l.k;
    hipLaunchKernelGGL(ct_flip_kernel, dim3((n + 255) / 256), dim3(256), 0, s, h->params + l.w_off, h->wflip, l.cout_p, l.cin_p, l.k);
    return launch_gemm_epi(dY, h->wflip, nullptr, dX, M, l.cin_p, l.k * l.cout_p, 0, 0.0f, 0, l.cout_p, l.cin_p,
                           l.k == 3 ? h->L : 0, l.k == 3 ? l.cout_p : 0, e, s);
}

// loss of the last forward against y_true + all gradients into grads (flat, same layout as the parameters; overwritten).
// grad_scale multiplies dLoss/dy (1/world_size shares for data-parallel training are applied by the caller through it).
extern "C" int csa_cnn_train_backward(csa_cnn_trainer *h, const float *y_true, float grad_scale, float *loss_out, float *grads, void *stream)
{
    if (!h || !y_true || !grads || h->last_B <= 0) { csa_set_error_msg("csa_cnn_train_backward: bad argument / no forward"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int B = h->last_B, M = B * h->L, wp = h->wp, cp = h->cout_p;
    const float gs = 1.0f / (1.0f - h->dropout);
    int rc;
    CSA_HIP_CHECK(hipMemsetAsync(grads, 0, sizeof(float) * h->n_params, s));
    const int nrest = h->cout - h->n_lin;
    const float w_lin = (120.0f / 128.0f) / ((float)M * h->n_lin), w_rest = (8.0f / 128.0f) / ((float)M * nrest);
    const int nb = (M + 255) / 256;
    hipLaunchKernelGGL(ct_loss_kernel, dim3(nb), dim3(256), 0, s, h->y, y_true, h->dy, h->lpart, M, h->cout, h->n_lin,
                       w_lin * grad_scale, w_rest * grad_scale);
    if (loss_out) hipLaunchKernelGGL(ct_loss_final_kernel, dim3(1), dim3(256), 0, s, h->lpart, nb, loss_out);
    const CtLayer &po = h->layers[3 * h->depth], &de = h->layers[3 * h->depth + 1];
    hipLaunchKernelGGL(ct_tail_bwd_kernel, dim3(nb), dim3(256), 0, s, h->dy, h->y, h->z, h->params + de.w_off, h->ddp, h->dzp, M, h->cout, cp, h->n_lin);
    if ((rc = ct_wgrad(h, de, h->ddp, h->z, M, 0, grads, s))) return rc;
    const float *hl = h->hb[h->depth - 1];
    if ((rc = ct_wgrad(h, po, h->dzp, hl, M, 0, grads, s))) return rc;
    float *dh = h->g0, *dt = h->g1, *dn = h->g2;
    if ((rc = ct_dgrad(h, po, h->dzp, dh, M, GemmEpi{}, s))) return rc;
    for (int i = h->depth - 1; i >= 0; --i) {
        const CtLayer &a = h->layers[3 * i], &b = h->layers[3 * i + 1], &r = h->layers[3 * i + 2];
        const float *in = i == 0 ? h->xin : h->hb[i - 1];
        const size_t n = (size_t)M * wp;
        // residual 1x1 projection
        if ((rc = ct_wgrad(h, r, dh, in, M, 0, grads, s))) return rc;
        if (i > 0 && (rc = ct_dgrad(h, r, dh, dn, M, GemmEpi{}, s))) return rc;
        // second conv: dt = dh gated by the saved t2, then its weight gradient and the input gradient gated by t1
        hipLaunchKernelGGL(ct_gate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dh, h->t2[i], dt, n, gs);
        if ((rc = ct_wgrad(h, b, dt, h->t1[i], M, 1, grads, s))) return rc;
        GemmEpi eg{};
        eg.gate = h->t1[i]; eg.gscale = gs;
        if ((rc = ct_dgrad(h, b, dt, dh, M, eg, s))) return rc;          // dh now holds d(pre-activation of conv_a)
        if ((rc = ct_wgrad(h, a, dh, in, M, 1, grads, s))) return rc;
        if (i > 0) {
            GemmEpi ea{};
            ea.addsrc = dn;
            if ((rc = ct_dgrad(h, a, dh, dt, M, ea, s))) return rc;       // d(block input) = conv_a path + residual path
            float *t = dh; dh = dt; dt = t;
        }
    }
    return CSA_OK;
}

extern "C" int csa_cnn_train_adam(csa_cnn_trainer *h, const float *grads, float lr, float beta1, float beta2, float eps, int step, void *stream)
{
    if (!h || !grads || step <= 0) { csa_set_error_msg("csa_cnn_train_adam: bad argument"); return CSA_ERR_ARG; }
    const float lr_t = lr * sqrtf(1.0f - powf(beta2, (float)step)) / (1.0f - powf(beta1, (float)step));
    hipLaunchKernelGGL(ct_adam_kernel, dim3((unsigned)((h->n_params + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->params, grads,
                       h->m1, h->m2, h->n_params, lr_t, beta1, beta2, eps);
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
