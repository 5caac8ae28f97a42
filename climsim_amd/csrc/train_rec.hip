// train_rec.hip -- backward (BPTT over the 60 levels) of the level-recurrent LSTM.
//
// Reference: autograd through nn.LSTM in the TBPTT loop, rnn/utils.py:1098-1137 (forward with
// graph) and :1366-1371 (loss.backward); cell arithmetic = PyTorch LSTM (gate order i,f,g,o).
//
// Same MI355X mapping as the forward kernel (rec.hip): W_hh^T (128 x 512 fp32 = 256 KB) stays in the
// VGPRs of one 512-thread workgroup for all 60 steps, two columns advance per workgroup with
// v_pk_fma_f32, one LDS-only barrier per step, no inter-workgroup traffic.
//
// Per step t = L-1 .. 0 and column:
//   dh      = dH_ext[t] + W_hh^T dp[t+1]                      (recurrent matvec, 512 -> 128)
//   do~     = dh * tanh(c_t) ;  dc = dh * o * (1 - tanh(c_t)^2) + dc_carry
//   di~ = dc*g ; dg~ = dc*i ; df~ = dc*c_{t-1} ; dc_carry = dc*f
//   dp[t]   = (di~ i(1-i), dg~ (1-g^2), df~ f(1-f), do~ o(1-o))   written IN PLACE over the saved gates
// dp (L,B,4*nh), unit-major [i,g~,f,o], then feeds the weight-gradient and input-gradient GEMMs.
//
// Thread (og, rc): og = tid>>4 owns the four outputs k = 4*og..4*og+3, rc = tid&15 the 32 rows
// r' = u*4+pos of chunk rc; 4*32 = 128 weights per lane.  The 16 row-chunks of an output sit in one
// DPP row: quad reduce-scatter (6 adds, same slot/column-swap trick as the forward kernel) followed by a
// two-step rotate all-reduce across the four quads.  Lane (og, q4, rcq<2) then owns the cell
// (u = 4*og + 2*(q4>>1) + rcq, col = q4&1): it keeps dc_carry in a register and does the elementwise
// gate gradient.
#include "common.h"

#define PK_FMA_LO(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(h))
#define PK_FMA_HI(acc, w, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(h))
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ float tanh_acc(float x)
{
    const float t = fminf(__builtin_amdgcn_exp2f(-2.88539008177792681f * x), 1e30f);
    return (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
}

// NL4 > 0 (nh = 144): as in rec.hip, 9 waves cap a wave at 168 VGPRs, so the last NL4 float4 of every slot's
// weight run live in (dynamic) LDS and are read once per step; the rest stays register-stationary.
template <int NH, int NL4 = 0>
__global__ __launch_bounds__(NH * 4, (NH / 16 + 3) / 4) void lstm_bwd_rec_kernel(
    const f32x4 *__restrict__ WTp4, float *__restrict__ GP, const float *__restrict__ Cseq,
    const float *__restrict__ dH, float *__restrict__ dh0, float *__restrict__ dc0,
    int B, int L, int rev)
{
    constexpr int NT = NH * 4;
    constexpr int RC = NH / 4;          // rows per chunk (16 chunks cover the 4*NH rows)
    constexpr int CH = 2 * RC + 4;      // floats per chunk incl. one 16-B pad slot
    constexpr int CPY = 16 * CH;
    static_assert(RC % 4 == 0, "nh must be a multiple of 16");
    constexpr int KR = RC / 2 - 2 * NL4;   // weight pairs per slot kept in registers
    __shared__ __attribute__((aligned(16))) float dpbuf[2][2 * CPY];
    extern __shared__ f32x4 wlds[];        // NL4 > 0: 4*NL4*NT float4

    const int tid = threadIdx.x, og = tid >> 4, rc = tid & 15, q4 = rc & 3, rcq = rc >> 2;
    const int col = q4 & 1, grp = q4 >> 1;
    int b = 2 * blockIdx.x + col;
    const bool valid = b < B;
    if (!valid) b = B - 1;
    const bool cell = rcq < 2;                       // this lane owns one (u, col) cell
    const int u = 4 * og + 2 * grp + (rcq & 1);

    f32x2 w[4][KR > 0 ? KR : 1];
#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const f32x4 v = WTp4[(size_t)i * NT + tid];
        const int s = (4 * i) / RC, kk = (4 * i) % RC;
        if (kk / 2 < KR) {
            w[s][kk / 2] = f32x2{v.x, v.y};
            w[s][kk / 2 + 1] = f32x2{v.z, v.w};
        } else {
            wlds[(s * NL4 + (kk / 2 - KR) / 2) * NT + tid] = v;
        }
    }
    // acc[s] += sum over this lane's rows of W^T * dp   (both columns), dp read from the LDS copy at dpp
#define BWD_MATVEC(acc, dpp)                                                                       \
    {                                                                                              \
        _Pragma("unroll") for (int j = 0; j < KR; ++j) {                                           \
            const f32x4 v = (dpp)[j];                                                              \
            const f32x2 va = {v.x, v.y}, vb = {v.z, v.w};                                          \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) PK_FMA_LO(acc[s], w[s][j], va);          \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) PK_FMA_HI(acc[s], w[s][j], vb);          \
        }                                                                                          \
        _Pragma("unroll") for (int q = 0; q < NL4; ++q) {                                          \
            const f32x4 v0 = (dpp)[KR + 2 * q], v1 = (dpp)[KR + 2 * q + 1];                        \
            const f32x2 va0 = {v0.x, v0.y}, vb0 = {v0.z, v0.w}, va1 = {v1.x, v1.y}, vb1 = {v1.z, v1.w}; \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                        \
                const f32x4 wv = wlds[(s * NL4 + q) * NT + tid];                                   \
                const f32x2 w0 = {wv.x, wv.y}, w1 = {wv.z, wv.w};                                  \
                PK_FMA_LO(acc[s], w0, va0);                                                        \
                PK_FMA_HI(acc[s], w0, vb0);                                                        \
                PK_FMA_LO(acc[s], w1, va1);                                                        \
                PK_FMA_HI(acc[s], w1, vb1);                                                        \
            }                                                                                      \
        }                                                                                          \
    }

    // LDS slots of this cell's four dp rows r' = u*4 + pos (normal and column-swapped copy)
    const int r0 = u * 4;
    const int cslot = (r0 / RC) * CH + (r0 % RC) * 2;      // 4 consecutive rows never straddle a chunk
    const int rdoff = col * CPY + rc * CH;

    float dc_carry = 0.0f, dh_rec = 0.0f;
    const size_t cidx = (size_t)b * NH + u;
    const size_t cstep = (size_t)B * NH;

    // Operands of the elementwise part (saved gates, c_{t-1}, external dh: none depends on the recurrence) are fetched ONE STEP
    // AHEAD with asm global loads into the other of two register sets and become valid behind `s_waitcnt vmcnt(3)` at the top of
    // the step that uses them, exactly the scheme of the forward kernels (rec.hip): hipcc knows nothing of a load in flight, so
    // it places no wait of its own.  Written as plain C++ loads (round 2) the compiler put `s_waitcnt vmcnt(0)` right behind the
    // three loads it had just issued -- every step paid a full memory round trip plus the ack of the previous step's 16-byte dp
    // store (SQ_WAIT_ANY 55 % of wave time, profiles/r3_train_tbptt3_384_sq_pmc_before.json).  In-order VMEM accounting per step:
    // [3 loads for t-1] [dp store of t]; vmcnt(3) at the top of step t-1 leaves only that step's own three loads outstanding, so
    // the store of step t has had a whole matvec to complete.  The prefetch is unconditional (step 0 re-reads its own row) and
    // there is ONE wait statement per step with ONE count (tools/check_asm_prefetch.py lints the emitted ISA).
    f32x4 g4A = {0, 0, 0, 0}, g4B = {0, 0, 0, 0};
    float cpA = 0.f, cpB = 0.f, dheA = 0.f, dheB = 0.f, c_t = 0.f;
    const float *GPu = GP + (size_t)b * (4 * NH) + u * 4;
    const size_t GPstep = (size_t)B * (4 * NH);
#define BWD_PREFETCH(TN, G4N, CPN, DHN)                                                                       \
    {                                                                                                         \
        const int tn_ = (TN);                                                                                 \
        const float *pg = GPu + (size_t)tn_ * GPstep, *pc = Cseq + (size_t)tn_ * cstep + cidx;                \
        const float *pd = dH + ((size_t)(rev ? L - 1 - tn_ : tn_) * B + b) * NH + u;                          \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(G4N) : "v"(pg) : "memory");                    \
        asm volatile("global_load_dword %0, %1, off" : "=&v"(CPN) : "v"(pc) : "memory");                      \
        asm volatile("global_load_dword %0, %1, off" : "=&v"(DHN) : "v"(pd) : "memory");                      \
    }
    if (cell) c_t = Cseq[(size_t)L * cstep + cidx];
    BWD_PREFETCH(L - 1, g4A, cpA, dheA)

    // one step: prefetch t-1, gate gradient of t (cell lanes), barrier, recurrent matvec on dp[t] -> dh_rec for step t-1
    // (after step 0: dh_rec = W_hh^T dp[0] = the gradient w.r.t. the initial hidden state)
    // PF = 1: the paired steps of the loop (prefetch + vmcnt(3)); PF = 0: the unpaired last step of an odd L -- its prefetch would
    // land in registers that are dead by then and that hipcc is free to re-use while the load is still in flight
#define BWD_STEP(PF, T, G4C, CPC, DHC, G4N, CPN, DHN)                                                         \
    {                                                                                                         \
        const int t_ = (T), cur = t_ & 1;                                                                     \
        if (PF) {                                                                                             \
            BWD_PREFETCH(t_ > 0 ? t_ - 1 : 0, G4N, CPN, DHN)                                                  \
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(G4C), "+v"(CPC), "+v"(DHC));                             \
        } else {                                                                                              \
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(G4C), "+v"(CPC), "+v"(DHC));                             \
        }                                                                                                     \
        if (cell) {                                                                                           \
            const f32x4 g4 = G4C;                                                                             \
            const float dh = DHC + dh_rec;                                                                    \
            const float tc = tanh_acc(c_t);                                                                   \
            const float dc = dh * g4.w * (1.0f - tc * tc) + dc_carry;                                         \
            f32x4 dp;                                                                                         \
            dp.x = dc * g4.y * g4.x * (1.0f - g4.x);             /* i  */                                     \
            dp.y = dc * g4.x * (1.0f - g4.y * g4.y);             /* g~ */                                     \
            dp.z = dc * CPC * g4.z * (1.0f - g4.z);              /* f  */                                     \
            dp.w = dh * tc * g4.w * (1.0f - g4.w);               /* o  */                                     \
            dc_carry = dc * g4.z;                                                                             \
            c_t = CPC;                                                                                        \
            float *n = &dpbuf[cur][cslot + col], *sw = &dpbuf[cur][CPY + cslot + (1 - col)];                  \
            n[0] = dp.x; n[2] = dp.y; n[4] = dp.z; n[6] = dp.w;                                               \
            sw[0] = dp.x; sw[2] = dp.y; sw[4] = dp.z; sw[6] = dp.w;                                           \
            if (valid) *(f32x4 *)(GPu + (size_t)t_ * GPstep) = dp;                                            \
        }                                                                                                     \
        LDS_BARRIER();                                                                                        \
        f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};                                      \
        const f32x4 *dpp = (const f32x4 *)&dpbuf[cur][rdoff];                                                 \
        BWD_MATVEC(acc, dpp)                                                                                  \
        float r[4];                                                                                           \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) r[s] = acc[s].x + dpp_mov<0xB1>(acc[s].y); /* xor 1 */  \
        float v0 = r[0] + dpp_mov<0x4E>(r[2]);                                                   /* xor 2 */  \
        float v1 = r[1] + dpp_mov<0x4E>(r[3]);                                                                \
        v0 += dpp_mov<0x124>(v0); v1 += dpp_mov<0x124>(v1);                                      /* row_ror:4 */ \
        v0 += dpp_mov<0x128>(v0); v1 += dpp_mov<0x128>(v1);                                      /* row_ror:8 */ \
        dh_rec = (rcq & 1) ? v1 : v0;                                                                         \
    }
    int t = L - 1;
    for (; t >= 1; t -= 2) {
        BWD_STEP(1, t, g4A, cpA, dheA, g4B, cpB, dheB)
        BWD_STEP(1, t - 1, g4B, cpB, dheB, g4A, cpA, dheA)
    }
    if (t == 0) BWD_STEP(0, 0, g4A, cpA, dheA, g4B, cpB, dheB)
#undef BWD_STEP
#undef BWD_PREFETCH
    // the last (unused) prefetch of an even L must land before its registers may be re-used: the operands keep them live until here
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(g4A), "+v"(cpA), "+v"(dheA), "+v"(g4B), "+v"(cpB), "+v"(dheB) : : "memory");
    if (cell && valid) {      // gradient w.r.t. the initial state: dh_init = W_hh^T dp[0], dc_init = dc_carry
        dh0[cidx] = dh_rec;
        dc0[cidx] = dc_carry;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// GRU backward (autograd through nn.GRU: r,z,n gates, n = tanh(x_n + r*(W_hn h + b_hn)), h' = (1-z) n + z h).
// Same mapping: W_hh^T (nh x 3nh) register-stationary, rows r' = u*3 + [r, z, hn] in 16 chunks, two columns per
// workgroup.  Saved per (t, b, u): [r, z, n, hn] (written by the TRAIN forward over the 4-padded pre-activations).
//   dh   = dH_ext[t] + dh_rec;   dn = dh (1-z);  dz = dh (h_{t-1} - n);  dh_direct = dh z
//   dn~  = dn (1-n^2);  g_hn = dn~ r;  dr~ = dn~ hn r(1-r);  dz~ = dz z(1-z)
//   dh_rec(t-1) = W_hh^T [dr~, dz~, g_hn] + dh_direct
// [dr~, dz~, dn~, g_hn] overwrites the saved gates in place: columns 0,1,2 are the gradient w.r.t. the input
// projection (dW_ih, dX, db_ih), columns 0,1,3 the one w.r.t. the recurrent projection (dW_hh, db_hh) -- the
// weight-gradient GEMMs run on all four columns and the scatter maps drop the unused one.
template <int NH>
__global__ __launch_bounds__(NH * 4, 2) void gru_bwd_rec_kernel(
    const f32x4 *__restrict__ WTp4, float *__restrict__ GP, const float *__restrict__ Hseq,
    const float *__restrict__ dH, float *__restrict__ dh0, int B, int L, int rev)
{
    constexpr int NT = NH * 4;
    constexpr int RC = 3 * NH / 16;     // rows per chunk
    constexpr int CH = 2 * RC + 4;
    constexpr int CPY = 16 * CH;
    static_assert(RC % 4 == 0 && RC % 3 == 0, "nh must be a multiple of 64");
    __shared__ __attribute__((aligned(16))) float dpbuf[2][2 * CPY];

    const int tid = threadIdx.x, og = tid >> 4, rc = tid & 15, q4 = rc & 3, rcq = rc >> 2;
    const int col = q4 & 1, grp = q4 >> 1;
    int b = 2 * blockIdx.x + col;
    const bool valid = b < B;
    if (!valid) b = B - 1;
    const bool cell = rcq < 2;
    const int u = 4 * og + 2 * grp + (rcq & 1);

    f32x2 w[4][RC / 2];
#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const f32x4 v = WTp4[(size_t)i * NT + tid];
        const int s = (4 * i) / RC, kk = (4 * i) % RC;
        w[s][kk / 2] = f32x2{v.x, v.y};
        w[s][kk / 2 + 1] = f32x2{v.z, v.w};
    }
    const int r0 = u * 3;
    const int cslot = (r0 / RC) * CH + (r0 % RC) * 2;
    const int rdoff = col * CPY + rc * CH;
    float dh_rec = 0.0f, dh_dir = 0.0f;
    const size_t cidx = (size_t)b * NH + u, cstep = (size_t)B * NH;

    // operands of the gate gradient one step ahead through asm loads + ONE counted wait per step, as lstm_bwd_rec_kernel above
    f32x4 g4A = {0, 0, 0, 0}, g4B = {0, 0, 0, 0};
    float hpA = 0.f, hpB = 0.f, dheA = 0.f, dheB = 0.f;
    const float *GPu = GP + (size_t)b * (4 * NH) + u * 4;
    const size_t GPstep = (size_t)B * (4 * NH);
#define GRU_PREFETCH(TN, G4N, HPN, DHN)                                                                       \
    {                                                                                                         \
        const int tn_ = (TN);                                                                                 \
        const float *pg = GPu + (size_t)tn_ * GPstep, *ph = Hseq + (size_t)tn_ * cstep + cidx;                \
        const float *pd = dH + ((size_t)(rev ? L - 1 - tn_ : tn_) * B + b) * NH + u;                          \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(G4N) : "v"(pg) : "memory");                    \
        asm volatile("global_load_dword %0, %1, off" : "=&v"(HPN) : "v"(ph) : "memory");                      \
        asm volatile("global_load_dword %0, %1, off" : "=&v"(DHN) : "v"(pd) : "memory");                      \
    }
    GRU_PREFETCH(L - 1, g4A, hpA, dheA)
#define GRU_MATVEC(acc, dpp)                                                                       \
    _Pragma("unroll") for (int j = 0; j < RC / 2; ++j) {                                           \
        const f32x4 v = (dpp)[j];                                                                  \
        const f32x2 va = {v.x, v.y}, vb = {v.z, v.w};                                              \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) PK_FMA_LO(acc[s], w[s][j], va);              \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) PK_FMA_HI(acc[s], w[s][j], vb);              \
    }
#define GRU_BWD_STEP(PF, T, G4C, HPC, DHC, G4N, HPN, DHN)                                                     \
    {                                                                                                         \
        const int t_ = (T), cur = t_ & 1;                                                                     \
        if (PF) {                                                                                             \
            GRU_PREFETCH(t_ > 0 ? t_ - 1 : 0, G4N, HPN, DHN)                                                  \
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(G4C), "+v"(HPC), "+v"(DHC));                             \
        } else {                                                                                              \
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(G4C), "+v"(HPC), "+v"(DHC));                             \
        }                                                                                                     \
        if (cell) {                                                                                           \
            const float dh = DHC + dh_rec;                                                                    \
            const float r = G4C.x, z = G4C.y, n = G4C.z, hn = G4C.w;                                          \
            const float dn = dh * (1.0f - z), dz = dh * (HPC - n);                                            \
            dh_dir = dh * z;                                                                                  \
            f32x4 dp;                                                                                         \
            dp.z = dn * (1.0f - n * n);              /* dn~  (input-projection side) */                       \
            dp.w = dp.z * r;                         /* g_hn (recurrent side) */                              \
            dp.x = dp.z * hn * r * (1.0f - r);       /* dr~ */                                                \
            dp.y = dz * z * (1.0f - z);              /* dz~ */                                                \
            float *nn = &dpbuf[cur][cslot + col], *sw = &dpbuf[cur][CPY + cslot + (1 - col)];                 \
            nn[0] = dp.x; nn[2] = dp.y; nn[4] = dp.w;                                                         \
            sw[0] = dp.x; sw[2] = dp.y; sw[4] = dp.w;                                                         \
            if (valid) *(f32x4 *)(GPu + (size_t)t_ * GPstep) = dp;                                            \
        }                                                                                                     \
        LDS_BARRIER();                                                                                        \
        f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};                                      \
        const f32x4 *dpp = (const f32x4 *)&dpbuf[cur][rdoff];                                                 \
        GRU_MATVEC(acc, dpp)                                                                                  \
        float rr[4];                                                                                          \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) rr[s] = acc[s].x + dpp_mov<0xB1>(acc[s].y);             \
        float v0 = rr[0] + dpp_mov<0x4E>(rr[2]);                                                              \
        float v1 = rr[1] + dpp_mov<0x4E>(rr[3]);                                                              \
        v0 += dpp_mov<0x124>(v0); v1 += dpp_mov<0x124>(v1);                                                   \
        v0 += dpp_mov<0x128>(v0); v1 += dpp_mov<0x128>(v1);                                                   \
        dh_rec = ((rcq & 1) ? v1 : v0) + dh_dir;                                                              \
    }
    int t = L - 1;
    for (; t >= 1; t -= 2) {
        GRU_BWD_STEP(1, t, g4A, hpA, dheA, g4B, hpB, dheB)
        GRU_BWD_STEP(1, t - 1, g4B, hpB, dheB, g4A, hpA, dheA)
    }
    if (t == 0) GRU_BWD_STEP(0, 0, g4A, hpA, dheA, g4B, hpB, dheB)
#undef GRU_BWD_STEP
#undef GRU_PREFETCH
#undef GRU_MATVEC
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(g4A), "+v"(hpA), "+v"(dheA), "+v"(g4B), "+v"(hpB), "+v"(dheB) : : "memory");
    if (cell && valid) dh0[cidx] = dh_rec;       // gradient w.r.t. the initial hidden state
}

size_t bwd_rec_packed_floats_gru(int nh) { return (size_t)3 * nh * nh; }
void bwd_rec_pack_weights_gru(int nh, const float *w_hh, float *packed)
{
    const int NT = nh * 4, RC = 3 * nh / 16;
    for (int tid = 0; tid < NT; ++tid) {
        const int og = tid >> 4, rc = tid & 15, grp = (rc & 3) >> 1;
        for (int idx = 0; idx < 4 * RC; ++idx) {
            const int s = idx / RC, rr = idx % RC;
            const int k = 4 * og + (s + 2 * grp) % 4;
            const int rp = rc * RC + rr;                    // r' = u*3 + pos, pos over [r, z, hn]
            const int uu = rp / 3, pos = rp % 3;
            const int i = idx / 4, e = idx % 4;
            packed[((size_t)i * NT + tid) * 4 + e] = w_hh[(size_t)(pos * nh + uu) * nh + k];
        }
    }
}
int launch_bwd_rec_gru(int nh, const float *wt_packed, float *GP, const float *Hseq, const float *dH, float *dh0, int B, int L,
                       int rev, hipStream_t s)
{
    const dim3 grid((B + 1) / 2), block(nh * 4);
    switch (nh) {
    case 64:  hipLaunchKernelGGL((gru_bwd_rec_kernel<64>), grid, block, 0, s, (const f32x4 *)wt_packed, GP, Hseq, dH, dh0, B, L, rev); break;
    case 128: hipLaunchKernelGGL((gru_bwd_rec_kernel<128>), grid, block, 0, s, (const f32x4 *)wt_packed, GP, Hseq, dH, dh0, B, L, rev); break;
    default:
        csa_set_error_msg("bwd_rec(GRU): hidden size not supported (64, 128)");
        return CSA_ERR_UNSUPPORTED;
    }
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}

// slot s of a lane with group grp accumulates output j = (s + 2*grp) % 4 of its quad of outputs
// row order inside a chunk: r' = u*4 + pos, pos over [i, g~, f, o]  ->  PyTorch gate row
static const int kPosGate[4] = {0, 2, 1, 3};

size_t bwd_rec_packed_floats(int nh) { return (size_t)4 * nh * nh; }

void bwd_rec_pack_weights(int nh, const float *w_hh, float *packed)
{
    const int NT = nh * 4, RC = nh / 4;
    for (int tid = 0; tid < NT; ++tid) {
        const int og = tid >> 4, rc = tid & 15, grp = (rc & 3) >> 1;
        for (int idx = 0; idx < 4 * RC; ++idx) {
            const int s = idx / RC, rr = idx % RC;
            const int k = 4 * og + (s + 2 * grp) % 4;
            const int rp = rc * RC + rr;                    // r' = u*4 + pos
            const int uu = rp / 4, pos = rp % 4;
            const int i = idx / 4, e = idx % 4;
            packed[((size_t)i * NT + tid) * 4 + e] = w_hh[(size_t)(kPosGate[pos] * nh + uu) * nh + k];
        }
    }
}

int launch_bwd_rec(int nh, const float *wt_packed, float *GP, const float *Cseq, const float *dH,
                   float *dh0, float *dc0, int B, int L, int rev, hipStream_t s)
{
    const dim3 grid((B + 1) / 2), block(nh * 4);
    switch (nh) {
    case 64:  hipLaunchKernelGGL((lstm_bwd_rec_kernel<64>), grid, block, 0, s, (const f32x4 *)wt_packed, GP, Cseq, dH, dh0, dc0, B, L, rev); break;
    case 96:  hipLaunchKernelGGL((lstm_bwd_rec_kernel<96>), grid, block, 0, s, (const f32x4 *)wt_packed, GP, Cseq, dH, dh0, dc0, B, L, rev); break;
    case 128: hipLaunchKernelGGL((lstm_bwd_rec_kernel<128>), grid, block, 0, s, (const f32x4 *)wt_packed, GP, Cseq, dH, dh0, dc0, B, L, rev); break;
    case 144: {
        constexpr int NL4 = 2;
        constexpr size_t shm = (size_t)4 * NL4 * 144 * 4 * sizeof(f32x4);
        auto kern = lstm_bwd_rec_kernel<144, NL4>;
        CSA_SET_DYN_LDS_ONCE(kern, shm);
        hipLaunchKernelGGL(kern, grid, block, shm, s, (const f32x4 *)wt_packed, GP, Cseq, dH, dh0, dc0, B, L, rev);
        break;
    }
    default:
        csa_set_error_msg("bwd_rec: hidden size not supported (64, 96, 128, 144)");
        return CSA_ERR_UNSUPPORTED;
    }
    CSA_HIP_CHECK(hipGetLastError());
    return CSA_OK;
}
