// stoch_bwd.hip -- backward (BPTT over the levels) of the reference's stochastic recurrent layers.
//
//   MyStochasticGRULayer5   the reference's hand-written backward: stochastic_gru_backward_elem_kernel
//                           rnn/models_torch_kernels.py:85-126, the reverse C++ sequence loop :176-232 (per level: an elementwise
//                           launch, three cuBLAS GEMMs for the recurrent gradients and three weight-gradient GEMMs), wrapped as
//                           FusedCUDAStochasticGRUSequence.backward :826-841.
//   MyStochasticLSTMLayer4  autograd through :1494-1531 (no native backward upstream).
//
// MI355X version, the mirror image of stoch.hip: the TRANSPOSED recurrent matrices stay in the registers of one 512-thread
// workgroup for all levels (160 weights per lane), two columns per workgroup, packed FMAs, LDS-only barriers: ONE launch walks
// the sequence backwards and leaves, in place of the saved activations, the per-level pre-activation gradients; every
// weight gradient is then ONE split-M TN GEMM over all T*B rows (the reference accumulates T small GEMMs per weight) and the
// input gradient one NT GEMM -- the kernels of the deterministic training step (train_misc.hip, gemm.hip).
//
// Thread (k, p): k = tid>>2 an output unit of the transposed matvec, p = tid&3 a quarter of the contraction rows; lanes p<2 own
// the cell (unit k, column p) for the elementwise part and keep its carried gradients in registers.
#include "common.h"
#include "stoch.h"
#include "train.h"

#define PK_FMA_LO(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(h))
#define PK_FMA_HI(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(h))
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

namespace {

__device__ __forceinline__ float q_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    return v;
}
__device__ __forceinline__ float q_pick(f32x2 a, int col)      // both components reduced by ALL lanes, then the lane's column
{
    const float sx = q_sum(a.x), sy = q_sum(a.y);
    return col ? sy : sx;
}
__device__ __forceinline__ float b_tanh(float x)
{
    const float t = fminf(__builtin_amdgcn_exp2f(-2.88539008177792681f * x), 1e30f);
    return (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
}

// out[k] (both columns) = sum over this lane's RQ rows of W^T[k][r] * g[r]; g = (row, column) pairs in LDS; four interleaved
// accumulator chains
template <int RQ>
__device__ __forceinline__ f32x2 t_matvec(const f32x2 (&w)[RQ / 2], const float *gbuf)
{
    const f32x4 *gp = (const f32x4 *)gbuf;
    f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < RQ / 2; ++j) {
        const f32x4 gv = gp[j];
        const f32x2 ga = {gv.x, gv.y}, gb = {gv.z, gv.w};
        PK_FMA_LO(acc[j & 3], w[j], ga);
        PK_FMA_HI(acc[j & 3], w[j], gb);
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
template <int RQ>
__device__ __forceinline__ void load_t(const f32x4 *Wp4, int NT, int tid, f32x2 (&w)[RQ / 2])
{
#pragma unroll
    for (int i = 0; i < RQ / 4; ++i) {
        const f32x4 v = Wp4[(size_t)i * NT + tid];
        w[2 * i] = f32x2{v.x, v.y};
        w[2 * i + 1] = f32x2{v.z, v.w};
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// LSTM4.  Saved by the TRAIN forward: A (T,B,5H) = [o, ex, ig, fg, gg] gate-major, Cseq / Hseq (T+1 slots).  Per level, backwards:
//   dh = d_out[t] + dh_rec (+ d_hT at the last level);  tc = tanh(c_t);  do = dh tc;  dc = dc_carry + dh o (1 - tc^2)
//   dz = do o (1-o)  [z = mean + eps ex]:  d_mean = dz;  d_logvar = dz eps ex / 2;  d_eps = dz ex
//   d_i = dc gg ig(1-ig);  d_f = dc c_{t-1} fg(1-fg);  d_g = dc ig (1-gg^2);  dc_carry = dc fg
//   dS[t] = [d_mean, d_logvar, d_i, d_f, d_g] overwrites A[t];  dh_rec = W_h^T-matvec(dS[t])   (W_h = weight_encoder[nx:], (H,5H))
template <int NH>
__global__ __launch_bounds__(NH * 4, 2) void stoch_lstm4_bwd_kernel(
    const f32x4 *__restrict__ WT4, float *__restrict__ A, const float *__restrict__ Cseq, const float *__restrict__ eps,
    const float *__restrict__ d_out, const float *__restrict__ d_hT, const float *__restrict__ d_cT,
    float *__restrict__ d_h0, float *__restrict__ d_c0, float *__restrict__ d_eps, int B, int T)
{
    constexpr int NT = NH * 4, RQ = 5 * NH / 4, CH = 2 * RQ + 4;
    __shared__ __attribute__((aligned(16))) float gbuf[2][4 * CH];
    const int tid = threadIdx.x, k = tid >> 2, p = tid & 3, col = p & 1;
    int b = 2 * blockIdx.x + col;
    const bool valid = b < B;
    if (!valid) b = B - 1;
    const bool cell = p < 2;
    f32x2 w[RQ / 2];
    load_t<RQ>(WT4, NT, tid, w);
    float dh_rec = 0.0f, dc_carry = 0.0f;
    if (cell) {
        if (d_hT) dh_rec = d_hT[(size_t)b * NH + k];
        if (d_cT) dc_carry = d_cT[(size_t)b * NH + k];
    }
    // LDS slot of gradient row r = g*NH + k, column col: quarter r / RQ, position r % RQ
    int gslot[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) { const int r = g * NH + k; gslot[g] = (r / RQ) * CH + 2 * (r % RQ) + col; }
    for (int t = T - 1; t >= 0; --t) {
        const int cur = t & 1;
        if (cell) {
            const size_t row = (size_t)t * B + b;
            float *a = A + row * 5 * NH + k;
            const float o = a[0], ex = a[NH], ig = a[2 * NH], fg = a[3 * NH], gg = a[4 * NH];
            const float c_t = Cseq[((size_t)(t + 1) * B + b) * NH + k], c_p = Cseq[row * NH + k];
            const float e = eps[row * NH + k];
            const float dh = d_out[row * NH + k] + dh_rec;
            const float tc = b_tanh(c_t);
            const float dc = dc_carry + dh * o * (1.0f - tc * tc);
            const float dz = dh * tc * o * (1.0f - o);
            float dS[5];
            dS[0] = dz;
            dS[1] = dz * e * ex * 0.5f;
            dS[2] = dc * gg * ig * (1.0f - ig);
            dS[3] = dc * c_p * fg * (1.0f - fg);
            dS[4] = dc * ig * (1.0f - gg * gg);
            dc_carry = dc * fg;
#pragma unroll
            for (int g = 0; g < 5; ++g) {
                gbuf[cur][gslot[g]] = dS[g];
                if (valid) a[g * NH] = dS[g];
            }
            if (d_eps && valid) d_eps[row * NH + k] = dz * ex;
        }
        LDS_BARRIER();
        dh_rec = q_pick(t_matvec<RQ>(w, &gbuf[cur][p * CH]), col);
    }
    if (cell && valid) {
        d_h0[(size_t)b * NH + k] = dh_rec;
        d_c0[(size_t)b * NH + k] = dc_carry;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// GRU5 (models_torch_kernels.py:85-126, :195-230).  Saved by the TRAIN forward: (r, zg, n) over XP (T,B,3H), ZN, Zs, EX (T,B,H),
// Hseq (T+1 slots).  Per level, backwards:
//   dh = d_out[t] + dh_rec;  dn = dh (1-zg);  dzg = dh (h_{t-1} - n);  dh_elem = dh zg
//   dn~ = dn (1-n^2);  dr~ = dn~ z_n r(1-r);  dzg~ = dzg zg(1-zg);  dzn = dn~ r
//   gx[t]  = [dr~, dzg~, dn~]  (gradient of x W_ih + b_ih)   overwrites XP[t]
//   gzr[t] = [dr~, dzg~, dzn]  (gradient of z W_zh + b_zh)   -> GZ[t]
//   gz = W_zh^T-matvec(gzr);  d_eps = gz ex;  d_logvar = gz eps ex / 2;  gp[t] = [gz, d_logvar] -> GP[t]
//   dh_rec = dh_elem + W_enc^T-matvec(gp)
template <int NH>
__global__ __launch_bounds__(NH * 4, 2) void stoch_gru5_bwd_kernel(
    const f32x4 *__restrict__ WzT4, const f32x4 *__restrict__ WeT4, float *__restrict__ XP, const float *__restrict__ ZN,
    const float *__restrict__ EX, const float *__restrict__ Hseq, const float *__restrict__ eps, const float *__restrict__ d_out,
    float *__restrict__ GZ, float *__restrict__ GPd, float *__restrict__ d_h0, float *__restrict__ d_eps, int B, int T)
{
    constexpr int NT = NH * 4, RZ = 3 * NH / 4, RE = 2 * NH / 4, CHZ = 2 * RZ + 4, CHE = 2 * RE + 4;
    __shared__ __attribute__((aligned(16))) float gzbuf[4 * CHZ];
    __shared__ __attribute__((aligned(16))) float gpbuf[4 * CHE];
    const int tid = threadIdx.x, k = tid >> 2, p = tid & 3, col = p & 1;
    int b = 2 * blockIdx.x + col;
    const bool valid = b < B;
    if (!valid) b = B - 1;
    const bool cell = p < 2;
    f32x2 wz[RZ / 2], we[RE / 2];
    load_t<RZ>(WzT4, NT, tid, wz);
    load_t<RE>(WeT4, NT, tid, we);
    int zslot[3], eslot[2];
#pragma unroll
    for (int g = 0; g < 3; ++g) { const int r = g * NH + k; zslot[g] = (r / RZ) * CHZ + 2 * (r % RZ) + col; }
#pragma unroll
    for (int g = 0; g < 2; ++g) { const int r = g * NH + k; eslot[g] = (r / RE) * CHE + 2 * (r % RE) + col; }
    float dh_rec = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
        const size_t row = (size_t)t * B + b;
        float dh_elem = 0.0f, ex = 0.0f, e = 0.0f;
        if (cell) {
            float *a = XP + row * 3 * NH + k;
            const float r = a[0], zg = a[NH], n = a[2 * NH];
            const float zn = ZN[row * NH + k], hp = Hseq[row * NH + k];
            ex = EX[row * NH + k];
            e = eps[row * NH + k];
            const float dh = d_out[row * NH + k] + dh_rec;
            const float dn = dh * (1.0f - zg), dzg = dh * (hp - n);
            dh_elem = dh * zg;
            const float dn_pre = dn * (1.0f - n * n);
            const float dr_pre = dn_pre * zn * r * (1.0f - r);
            const float dzg_pre = dzg * zg * (1.0f - zg);
            const float dzn = dn_pre * r;
            gzbuf[zslot[0]] = dr_pre; gzbuf[zslot[1]] = dzg_pre; gzbuf[zslot[2]] = dzn;
            if (valid) {
                a[0] = dr_pre; a[NH] = dzg_pre; a[2 * NH] = dn_pre;
                float *gz = GZ + row * 3 * NH + k;
                gz[0] = dr_pre; gz[NH] = dzg_pre; gz[2 * NH] = dzn;
            }
        }
        LDS_BARRIER();
        const float gzv = q_pick(t_matvec<RZ>(wz, &gzbuf[p * CHZ]), col);
        if (cell) {
            const float glv = gzv * e * ex * 0.5f;
            gpbuf[eslot[0]] = gzv; gpbuf[eslot[1]] = glv;
            if (valid) {
                GPd[row * 2 * NH + k] = gzv; GPd[row * 2 * NH + NH + k] = glv;
                if (d_eps) d_eps[row * NH + k] = gzv * ex;
            }
        }
        LDS_BARRIER();
        dh_rec = dh_elem + q_pick(t_matvec<RE>(we, &gpbuf[p * CHE]), col);
    }
    if (cell && valid) d_h0[(size_t)b * NH + k] = dh_rec;
}

}  // namespace
// transposed packing: thread (k, p) holds W[k][p*RQ + j], j < RQ, W (nh, ncols) row-major in the reference's (in, out) layout
void stoch_pack_t(int nh, int ncols, const float *W, float *packed)
{
    const int NT = nh * 4, RQ = ncols / 4;
    for (int tid = 0; tid < NT; ++tid) {
        const int k = tid >> 2, p = tid & 3;
        for (int j = 0; j < RQ; ++j)
            packed[((size_t)(j / 4) * NT + tid) * 4 + (j % 4)] = W[(size_t)k * ncols + p * RQ + j];
    }
}
namespace {

float *dev_alloc(csa_stoch *h, size_t n, const float *src, int &rc)
{
    void *p = nullptr;
    if (hipMalloc(&p, sizeof(float) * (n ? n : 1)) != hipSuccess) { rc = CSA_ERR_NOMEM; return nullptr; }
    h->owned.push_back(p);
    if (src && hipMemcpy(p, src, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
    return (float *)p;
}

constexpr int kSplit = 64;

// grads[off : off + N1*N2] += A^T B over the M rows (deterministic two-pass reduction, no atomics)
int wgrad(csa_stoch *h, const float *A, int lda, const float *Bm, int ldb, int M, int N1, int N2, float *grads, hipStream_t s)
{
    int rc = launch_gemm_tn_partial(A, lda, Bm, ldb, h->part, M, N1, N2, kSplit, s);
    if (rc) return rc;
    return launch_reduce_partials(h->part, kSplit, N1 * N2, nullptr, nullptr, grads, s);
}
int bgrad(csa_stoch *h, const float *A, int M, int N, float *grads, hipStream_t s)
{
    int rc = launch_colsum_partial(A, h->part, M, N, kSplit, s);
    if (rc) return rc;
    return launch_reduce_partials(h->part, kSplit, N, nullptr, nullptr, grads, s);
}

}  // namespace

extern "C" int csa_stoch_enable_training(csa_stoch *h)
{
    if (!h) return CSA_ERR_ARG;
    if (h->Hseq) return CSA_OK;
    const int nh = h->nh;
    const size_t R = h->max_rows, R1 = 2 * R;        // (T+1)*B <= 2*T*B rows for the T+1-slot sequences
    int rc = CSA_OK;
    std::vector<float> pk;
    if (h->kind == 1) {
        pk.resize((size_t)nh * 5 * nh);
        stoch_pack_t(nh, 5 * nh, h->host_a.data(), pk.data());
        h->wT_a = dev_alloc(h, pk.size(), pk.data(), rc);
        h->Cseq = dev_alloc(h, R1 * nh, nullptr, rc);
    } else {
        pk.resize((size_t)nh * 3 * nh);
        stoch_pack_t(nh, 3 * nh, h->host_a.data(), pk.data());
        h->wT_a = dev_alloc(h, pk.size(), pk.data(), rc);
        pk.resize((size_t)nh * 2 * nh);
        stoch_pack_t(nh, 2 * nh, h->host_b.data(), pk.data());
        h->wT_b = dev_alloc(h, pk.size(), pk.data(), rc);
        h->ZN = dev_alloc(h, R * nh, nullptr, rc);
        h->Zs = dev_alloc(h, R * nh, nullptr, rc);
        h->EX = dev_alloc(h, R * nh, nullptr, rc);
        h->GZ = dev_alloc(h, R * 3 * nh, nullptr, rc);
        h->GPd = dev_alloc(h, R * 2 * nh, nullptr, rc);
    }
    const size_t widest = (size_t)(h->nx > nh ? h->nx : nh) * 5 * nh;
    h->part = dev_alloc(h, (size_t)kSplit * widest, nullptr, rc);
    float *hs = dev_alloc(h, R1 * nh, nullptr, rc);
    if (rc) return rc;
    h->Hseq = hs;
    h->own[0] = h->XP; h->own[1] = h->Hseq; h->own[2] = h->Cseq; h->own[3] = h->ZN; h->own[4] = h->Zs; h->own[5] = h->EX;
    return CSA_OK;
}

// sizes (floats) of the saved activations of one forward of T x B rows: XP | Hseq | Cseq (LSTM4)  or  XP | Hseq | ZN | Zs | EX (GRU5)
extern "C" long csa_stoch_activation_floats(const csa_stoch *h, int T, int B)
{
    if (!h || T <= 0 || B <= 0) return CSA_ERR_ARG;
    const long M = (long)T * B, M1 = (long)(T + 1) * B, nh = h->nh;
    return h->kind == 1 ? M * 5 * nh + 2 * M1 * nh : M * 3 * nh + M1 * nh + 3 * M * nh;
}

extern "C" int csa_stoch_set_activations(csa_stoch *h, float *acts, int T, int B)
{
    if (!h || !h->own[1]) { csa_set_error_msg("csa_stoch_set_activations: call csa_stoch_enable_training first"); return CSA_ERR_ARG; }
    if (!acts) {
        h->XP = h->own[0]; h->Hseq = h->own[1]; h->Cseq = h->own[2]; h->ZN = h->own[3]; h->Zs = h->own[4]; h->EX = h->own[5];
        return CSA_OK;
    }
    if (T <= 0 || B <= 0 || (long)T * B > h->max_rows) { csa_set_error_msg("csa_stoch_set_activations: bad shape"); return CSA_ERR_ARG; }
    const size_t M = (size_t)T * B, M1 = (size_t)(T + 1) * B, nh = h->nh;
    float *p = acts;
    h->XP = p; p += M * (h->kind == 1 ? 5 : 3) * nh;
    h->Hseq = p; p += M1 * nh;
    if (h->kind == 1) { h->Cseq = p; }
    else { h->ZN = p; p += M * nh; h->Zs = p; p += M * nh; h->EX = p; }
    return CSA_OK;
}

extern "C" long csa_stoch_num_params(const csa_stoch *h)
{
    if (!h) return CSA_ERR_ARG;
    const long nh = h->nh, nx = h->nx;
    return h->kind == 1 ? (nx + nh) * 5 * nh : nx * 3 * nh + nh * 3 * nh + nh * 2 * nh + (h->has_bias ? 6 * nh : 0);
}

// d_out (T,B,H); d_hT, d_cT (B,H) nullable -> d_x (T,B,nx), d_h0, d_c0 (B,H), d_eps (T,B,H) nullable;
// grads (flat, ACCUMULATED): weight_encoder ((nx+H), 5H) in the reference's layout
extern "C" int csa_stoch_lstm4_backward(csa_stoch *h, int T, int B, const float *x, const float *eps, const float *d_out,
                                        const float *d_hT, const float *d_cT, float *d_x, float *d_h0, float *d_c0, float *d_eps,
                                        float *grads, void *stream)
{
    if (!h || h->kind != 1 || !h->Hseq || !x || !eps || !d_out || !d_x || !d_h0 || !d_c0 || !grads || T <= 0 || B <= 0 ||
        (long)T * B > h->max_rows) { csa_set_error_msg("csa_stoch_lstm4_backward: bad argument (or training not enabled)"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int nh = h->nh, nx = h->nx, M = T * B;
    const dim3 grid((B + 1) / 2), block(nh * 4);
#define LB(NHv) hipLaunchKernelGGL((stoch_lstm4_bwd_kernel<NHv>), grid, block, 0, s, (const f32x4 *)h->wT_a, h->XP, h->Cseq, eps, d_out, \
                                   d_hT, d_cT, d_h0, d_c0, d_eps, B, T)
    switch (nh) {
    case 64: LB(64); break;
    case 96: LB(96); break;
    default: LB(128); break;
    }
#undef LB
    CSA_HIP_CHECK(hipGetLastError());
    int rc;
    // d_x = dS W_x^T: W_x = weight_encoder[:nx] is (nx, 5H) row-major = the (N, K) operand of the NT GEMM as it is
    if ((rc = launch_proj_gemm(h->XP, h->w_ref_in, nullptr, d_x, M, nx, 5 * nh, s))) return rc;
    if ((rc = wgrad(h, x, nx, h->XP, 5 * nh, M, nx, 5 * nh, grads, s))) return rc;
    return wgrad(h, h->Hseq, nh, h->XP, 5 * nh, M, nh, 5 * nh, grads + (size_t)nx * 5 * nh, s);
}

// grads (flat, ACCUMULATED): weight_ih (nx,3H) | weight_zh (H,3H) | weight_encoder (H,2H) [| bias_ih (3H) | bias_zh (3H)]
extern "C" int csa_stoch_gru5_backward(csa_stoch *h, int T, int B, const float *x, const float *eps, const float *d_out,
                                       float *d_x, float *d_h0, float *d_eps, float *grads, void *stream)
{
    if (!h || h->kind != 0 || !h->Hseq || !x || !eps || !d_out || !d_x || !d_h0 || !grads || T <= 0 || B <= 0 ||
        (long)T * B > h->max_rows) { csa_set_error_msg("csa_stoch_gru5_backward: bad argument (or training not enabled)"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int nh = h->nh, nx = h->nx, M = T * B;
    const dim3 grid((B + 1) / 2), block(nh * 4);
#define GB(NHv) hipLaunchKernelGGL((stoch_gru5_bwd_kernel<NHv>), grid, block, 0, s, (const f32x4 *)h->wT_a, (const f32x4 *)h->wT_b, h->XP, \
                                   h->ZN, h->EX, h->Hseq, eps, d_out, h->GZ, h->GPd, d_h0, d_eps, B, T)
    switch (nh) {
    case 64: GB(64); break;
    case 96: GB(96); break;
    default: GB(128); break;
    }
#undef GB
    CSA_HIP_CHECK(hipGetLastError());
    int rc;
    float *g_ih = grads, *g_zh = g_ih + (size_t)nx * 3 * nh, *g_enc = g_zh + (size_t)nh * 3 * nh, *g_bih = g_enc + (size_t)nh * 2 * nh;
    if ((rc = launch_proj_gemm(h->XP, h->w_ref_in, nullptr, d_x, M, nx, 3 * nh, s))) return rc;          // d_x = gx W_ih^T
    if ((rc = wgrad(h, x, nx, h->XP, 3 * nh, M, nx, 3 * nh, g_ih, s))) return rc;                          // x^T gx
    if ((rc = wgrad(h, h->Zs, nh, h->GZ, 3 * nh, M, nh, 3 * nh, g_zh, s))) return rc;                      // z^T gzr        (:213)
    if ((rc = wgrad(h, h->Hseq, nh, h->GPd, 2 * nh, M, nh, 2 * nh, g_enc, s))) return rc;                  // h_{t-1}^T gp  (:221)
    if (h->has_bias) {
        if ((rc = bgrad(h, h->XP, M, 3 * nh, g_bih, s))) return rc;
        if ((rc = bgrad(h, h->GZ, M, 3 * nh, g_bih + 3 * nh, s))) return rc;                                 // (:214)
    }
    return CSA_OK;
}
