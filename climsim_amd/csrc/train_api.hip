// train_api.hip -- C ABI of the training step (current-generation LSTM with memory):
//   csa_train_forward  : RNN_autoreg.forward in training mode, activations of time slot tau kept
//   csa_train_backward : backward of that slot (BPTT over the 60 levels x 2 LSTMs), gradients
//                        ACCUMULATED into one canonical flat fp32 buffer, d(mem_in) returned so the
//                        caller chains the slots of a TBPTT window (rnn/utils.py:1098-1137,1200-1377)
//   csa_train_loss     : huber + w_h*energy + w_w*water with its gradient (rnn/metrics.py)
//   csa_train_adam     : torch.optim.Adam arithmetic on the flat parameters, then re-pack
// Parameters live in ONE canonical flat device buffer in state_dict layout (csa_train_param_info);
// every kernel-specific layout (permuted W_ih, register-stationary W_hh / W_hh^T, transposed MLPs)
// is a device-side gather of it through index maps built once on the host -- by running the same
// host packers as the inference handle on arrays whose values are their own indices.
#include "common.h"
#include "pack.h"
#include "train.h"
#include "stoch.h"
#include <algorithm>
#include <string>
#include <vector>

namespace {

struct PInfo { std::string name; int off, rows, cols; };

struct Gather { float *dst; int *idx; int *idx2; int n; };

struct Slot {   // saved activations of one time step of the window
    float *X16, *xs, *X1, *GP1, *C1, *H1lev, *H1seq, *GP2, *C2, *H2, *Z, *hc0;
    // add_stochastic_layer: layer "1" = rnn0 (down), layer "2" = rnn1 (up), then the stochastic LSTM (down)
    float *Hb = nullptr;               // rnn1's hidden sequence in level order (input of the stochastic layer)
    float *Zs = nullptr;               // the stochastic layer's output (L,B,nh), the head's input
    float *sXP = nullptr, *sH = nullptr, *sC = nullptr;     // its saved activations (swapped into the csa_stoch handle per call)
    const float *eps = nullptr;        // the caller's noise of this slot (kept alive by the caller until backward)
};

}  // namespace

struct csa_trainer {
    DevModel dm;
    int max_batch, max_window, nparam;
    std::vector<PInfo> info;
    std::vector<void *> owned;
    float *params, *adam_m, *adam_v;
    float *hyai, *hybi;
    std::vector<Gather> gathers;
    GatherEntry *gtab = nullptr;     // device copy of the gather table (repack = one launch)
    int gmax = 0;
    float *wih1T, *wih2T, *whh1Tp, *whh2Tp;
    // gradient scatter maps
    int *map_wih1, *map_whh1, *map_b1a, *map_b1b, *map_wih2, *map_whh2, *map_b2a, *map_b2b, *map_head, *map_prep;
    std::vector<Slot> slots;
    // work buffers
    float *dH2, *dH1, *dX1, *dhc1, *dhc2, *part, *samp, *ecoef, *sp;
    float *part_b = nullptr;         // partial bias gradients (column sums of dP), a side output of the W_ih GEMM
    float *rtmp = nullptr;           // first-stage sums of the per-column partial reductions
    // one queued reduction per backward call / flush (train_misc.hip::reduce_partials_multi_kernel): every weight-gradient GEMM keeps
    // its partials in a region of the arena, the prep partials have a buffer (and first-stage sums) of their own
    float *arena = nullptr, *part2 = nullptr, *rtmp2 = nullptr;
    size_t arena_floats = 0;
    size_t part_floats;
    int nsplit;
    // deferred weight gradients (csa_train_set_deferred): the W_ih / W_hh gradient GEMMs of the backward calls are
    // postponed and done ONCE over all pending time steps of the window (csa_train_flush_wgrad)
    bool defer = false;
    std::vector<int> pending;
    int pending_B = 0;
    csa_stoch *stoch = nullptr;       // add_stochastic_layer: MyStochasticLSTMLayer4 (weights are gathers of the flat parameters)
    int off_enc = 0;                  // offset of rnn2.weight_encoder in the flat parameter / gradient buffers
    float *dscr = nullptr;            // discarded gradients of the noise initial state of rnn0
    float sp_scale = 1.f, sp_shift = 0.f;     // xdiv_sca[0], xmean_sca[0]: surface pressure de-normalisation of the loss
    // optional per-stage timing with HIP events on the call's own stream (csa_train_set_profiling; bench.py's roofline)
    bool profiling = false;
    struct EvPair { hipEvent_t a, b; int stage; };
    std::vector<EvPair> ev_pending;
    std::vector<hipEvent_t> ev_free;
    double acc_ms[CSA_TRAIN_NSTAGE] = {};
    long acc_n[CSA_TRAIN_NSTAGE] = {};
};

namespace {

template <typename T> T *dalloc(csa_trainer *h, size_t n, int &rc)
{
    void *p = nullptr;
    if (hipMalloc(&p, sizeof(T) * (n ? n : 1)) != hipSuccess) { rc = CSA_ERR_NOMEM; return nullptr; }
    h->owned.push_back(p);
    return (T *)p;
}
int *upload_idx(csa_trainer *h, const std::vector<int> &v, int &rc)
{
    int *d = dalloc<int>(h, v.size(), rc);
    if (d && hipMemcpy(d, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
    return d;
}
// register a gather: device buffer `n` floats filled from the flat params through idx (and idx2)
const float *add_gather(csa_trainer *h, const std::vector<int> &idx, const std::vector<int> *idx2, int &rc)
{
    Gather g;
    g.n = (int)idx.size();
    g.dst = dalloc<float>(h, idx.size(), rc);
    g.idx = upload_idx(h, idx, rc);
    g.idx2 = idx2 ? upload_idx(h, *idx2, rc) : nullptr;
    h->gathers.push_back(g);
    return g.dst;
}
std::vector<int> iota_off(int off, int n) { std::vector<int> v(n); for (int i = 0; i < n; ++i) v[i] = off + i; return v; }
std::vector<int> transposed_idx(int off, int O, int K)   // (O,K) source -> (K,O) buffer
{
    std::vector<int> v((size_t)O * K);
    for (int o = 0; o < O; ++o) for (int k = 0; k < K; ++k) v[(size_t)k * O + o] = off + o * K + k;
    return v;
}
std::vector<int> to_int(const std::vector<float> &f, int off)
{
    std::vector<int> v(f.size());
    for (size_t i = 0; i < f.size(); ++i) v[i] = off + (int)f[i];
    return v;
}
std::vector<float> index_values(size_t n) { std::vector<float> v(n); for (size_t i = 0; i < n; ++i) v[i] = (float)i; return v; }

// ---- stage timing: event pairs around single launches, collected lazily (host sync only in csa_train_get_profile) ----
hipEvent_t prof_event(csa_trainer *h)
{
    hipEvent_t e = nullptr;
    if (!h->ev_free.empty()) { e = h->ev_free.back(); h->ev_free.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void prof_collect(csa_trainer *h)
{
    for (auto &p : h->ev_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->acc_ms[p.stage] += ms;
            h->acc_n[p.stage] += 1;
        }
        h->ev_free.push_back(p.a);
        h->ev_free.push_back(p.b);
    }
    h->ev_pending.clear();
}
struct StageTimer {     // brackets the launches issued during its lifetime when profiling is on
    csa_trainer *h; hipStream_t s; int stage; hipEvent_t a = nullptr;
    StageTimer(csa_trainer *h_, int stage_, hipStream_t s_) : h(h_), s(s_), stage(stage_)
    {
        if (!h->profiling) return;
        if (h->ev_pending.size() >= 512) prof_collect(h);
        a = prof_event(h);
        if (a) (void)hipEventRecord(a, s);
    }
    ~StageTimer()
    {
        if (!a) return;
        hipEvent_t b = prof_event(h);
        if (!b) { h->ev_free.push_back(a); return; }
        (void)hipEventRecord(b, s);
        h->ev_pending.push_back({a, b, stage});
    }
};

int repack(csa_trainer *h, hipStream_t s)
{
    if (!h->gtab) {      // first call: the table of all gathers goes to the device once
        std::vector<GatherEntry> t;
        for (const Gather &g : h->gathers) { t.push_back({g.dst, g.idx, g.idx2, g.n}); h->gmax = g.n > h->gmax ? g.n : h->gmax; }
        void *d = nullptr;
        if (hipMalloc(&d, sizeof(GatherEntry) * t.size()) != hipSuccess) return CSA_ERR_NOMEM;
        h->owned.push_back(d);
        if (hipMemcpy(d, t.data(), sizeof(GatherEntry) * t.size(), hipMemcpyHostToDevice) != hipSuccess) return CSA_ERR_HIP;
        h->gtab = (GatherEntry *)d;
    }
    return launch_gather_multi(h->gtab, (int)h->gathers.size(), h->gmax, h->params, s);
}

}  // namespace

extern "C" int csa_train_create(const csa_config *cfg, const csa_params *p, const float *hyai, const float *hybi,
                                int max_batch, int max_window, csa_trainer **out)
{
    if (!cfg || !p || !hyai || !hybi || !out || max_batch <= 0 || max_window <= 0) { csa_set_error_msg("csa_train_create: bad argument"); return CSA_ERR_ARG; }
    const csa_config &c = *cfg;
    if (c.legacy || c.nh_mem <= 0 || (c.mp_mode != 1 && c.mp_mode != -1 && c.mp_mode != -2) || c.ny != (c.mp_mode == 1 ? 5 : 6)) {
        // mp_mode 0's training loop raises upstream (rnn/utils.py:1243 calls postprocessing with two arguments, models.py:274 takes three)
        csa_set_error_msg("csa_train_create: the HIP training step covers the current-generation LSTM / GRU with memory, mp_mode 1 (ny 5), -1 and -2 (ny 6)");
        return CSA_ERR_UNSUPPORTED;
    }
    if (c.add_stochastic_layer && (!c.use_lstm || c.nh1 != c.nh2 || c.nh1 > 128 || !p->rnn0_w_ih || !p->rnn2_weight_encoder)) {
        csa_set_error_msg("csa_train_create: add_stochastic_layer needs the LSTM flavour with nh1 == nh2 <= 128 and rnn0 / rnn2.weight_encoder");
        return CSA_ERR_UNSUPPORTED;
    }
    if (!c.use_lstm && !((c.nh1 == 64 || c.nh1 == 128) && (c.nh2 == 64 || c.nh2 == 128))) {
        csa_set_error_msg("csa_train_create: GRU training is built for hidden sizes 64 and 128");
        return CSA_ERR_UNSUPPORTED;
    }
    const bool ok = (c.nh1 == 64 || c.nh1 == 96 || c.nh1 == 128 || c.nh1 == 144) && (c.nh2 == 64 || c.nh2 == 96 || c.nh2 == 128 || c.nh2 == 144);
    if (!ok) { csa_set_error_msg("csa_train_create: hidden size must be 64, 96, 128 or 144"); return CSA_ERR_UNSUPPORTED; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { csa_set_error_msg("csa_train_create: no HIP device"); return CSA_ERR_HIP; }

    csa_trainer *h = new csa_trainer();
    memset(&h->dm, 0, sizeof(h->dm));
    h->dm.cfg = c;
    h->max_batch = max_batch;
    h->max_window = max_window;
    int rc = CSA_OK;
    const int L = c.nlev, nxp = c.nx + 1, nxs = c.nx_sfc, nh1 = c.nh1, nh2 = c.nh2, nm = c.nh_mem, nin1 = nh1 + nm;
    const int nhm = nh1 > nh2 ? nh1 : nh2;

    // ---- canonical flat layout (state_dict order of rnn/models/models.py::RNN_autoreg) ----------
    struct Src { const char *name; const float *ptr; int rows, cols; };
    const bool lstm = c.use_lstm != 0;
    const int G = lstm ? 4 : 3;                 // gate rows per unit in the state_dict (kernel layouts are padded to 4)
    std::vector<Src> srcs = {{"mlp_toa1.weight", p->mlp_toa1_w, nh2, 2}, {"mlp_toa1.bias", p->mlp_toa1_b, nh2, 1}};
    if (lstm) { srcs.push_back({"mlp_toa2.weight", p->mlp_toa2_w, nh2, 2}); srcs.push_back({"mlp_toa2.bias", p->mlp_toa2_b, nh2, 1}); }
    srcs.insert(srcs.end(), {{"mlp_initial.weight", p->mlp_initial_w, nh1, nxp}, {"mlp_initial.bias", p->mlp_initial_b, nh1, 1},
                             {"mlp_surface1.weight", p->mlp_surface1_w, nh1, nxs}, {"mlp_surface1.bias", p->mlp_surface1_b, nh1, 1}});
    if (lstm) { srcs.push_back({"mlp_surface2.weight", p->mlp_surface2_w, nh1, nxs}); srcs.push_back({"mlp_surface2.bias", p->mlp_surface2_b, nh1, 1}); }
    const bool st = c.add_stochastic_layer != 0;
    if (st)     // models.py:405-412: rnn0 (nh+nm -> nh), rnn1 (nh -> nh), rnn2 = MyStochasticLSTMLayer4(nh, nh)
        srcs.insert(srcs.end(), {
            {"rnn0.weight_ih_l0", p->rnn0_w_ih, G * nh1, nin1}, {"rnn0.weight_hh_l0", p->rnn0_w_hh, G * nh1, nh1},
            {"rnn0.bias_ih_l0", p->rnn0_b_ih, G * nh1, 1}, {"rnn0.bias_hh_l0", p->rnn0_b_hh, G * nh1, 1},
            {"rnn1.weight_ih_l0", p->rnn1_w_ih, G * nh2, nh1}, {"rnn1.weight_hh_l0", p->rnn1_w_hh, G * nh2, nh2},
            {"rnn1.bias_ih_l0", p->rnn1_b_ih, G * nh2, 1}, {"rnn1.bias_hh_l0", p->rnn1_b_hh, G * nh2, 1},
            {"rnn2.weight_encoder", p->rnn2_weight_encoder, nh1 + nh2, 5 * nh2}});
    else
        srcs.insert(srcs.end(), {
            {"rnn1.weight_ih_l0", p->rnn1_w_ih, G * nh1, nin1}, {"rnn1.weight_hh_l0", p->rnn1_w_hh, G * nh1, nh1},
            {"rnn1.bias_ih_l0", p->rnn1_b_ih, G * nh1, 1}, {"rnn1.bias_hh_l0", p->rnn1_b_hh, G * nh1, 1},
            {"rnn2.weight_ih_l0", p->rnn2_w_ih, G * nh2, nh1}, {"rnn2.weight_hh_l0", p->rnn2_w_hh, G * nh2, nh2},
            {"rnn2.bias_ih_l0", p->rnn2_b_ih, G * nh2, 1}, {"rnn2.bias_hh_l0", p->rnn2_b_hh, G * nh2, 1}});
    srcs.insert(srcs.end(), {
        {"mlp_latent.weight", p->mlp_latent_w, nm, nh2}, {"mlp_latent.bias", p->mlp_latent_b, nm, 1},
        {"mlp_output.weight", p->mlp_output_w, c.ny, nm}, {"mlp_output.bias", p->mlp_output_b, c.ny, 1},
        {"mlp_surface_output.weight", p->mlp_surface_output_w, c.ny_sfc, nh2},
        {"mlp_surface_output.bias", p->mlp_surface_output_b, c.ny_sfc, 1}});
    int off = 0;
    std::vector<float> flat;
    for (const Src &s : srcs) {
        if (!s.ptr) { delete h; csa_set_error_msg("csa_train_create: missing parameter array"); return CSA_ERR_ARG; }
        h->info.push_back({s.name, off, s.rows, s.cols});
        flat.insert(flat.end(), s.ptr, s.ptr + (size_t)s.rows * s.cols);
        off += s.rows * s.cols;
    }
    h->nparam = off;
    auto O = [&](const char *n) { for (auto &i : h->info) if (i.name == n) return i.off; return -1; };
    h->params = dalloc<float>(h, off, rc);
    h->adam_m = dalloc<float>(h, off, rc);
    h->adam_v = dalloc<float>(h, off, rc);
    if (rc == CSA_OK) {
        if (hipMemcpy(h->params, flat.data(), sizeof(float) * off, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
        if (hipMemset(h->adam_m, 0, sizeof(float) * off) != hipSuccess) rc = CSA_ERR_HIP;
        if (hipMemset(h->adam_v, 0, sizeof(float) * off) != hipSuccess) rc = CSA_ERR_HIP;
    }
    // constants
    auto upc = [&](const float *src, size_t n) { float *d = dalloc<float>(h, n, rc); if (d && hipMemcpy(d, src, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP; return d; };
    DevModel &d = h->dm;
    d.xmean_lev = upc(p->xmean_lev, (size_t)L * c.nx); d.xdiv_lev = upc(p->xdiv_lev, (size_t)L * c.nx);
    d.xmean_sca = upc(p->xmean_sca, nxs); d.xdiv_sca = upc(p->xdiv_sca, nxs);
    d.lbd_qc = upc(p->lbd_qc, L); d.lbd_qi = upc(p->lbd_qi, L);
    d.yscale_lev = upc(p->yscale_lev, (size_t)L * c.ny); d.yscale_sca = upc(p->yscale_sca, c.ny_sfc);
    d.hyam = upc(p->hyam, L); d.hybm = upc(p->hybm, L);
    h->hyai = upc(hyai, L + 1); h->hybi = upc(hybi, L + 1);
    h->sp_scale = p->xdiv_sca[0]; h->sp_shift = p->xmean_sca[0];

    // ---- gathers: canonical flat -> kernel layouts -------------------------------------------------
    d.toa1_wt = add_gather(h, transposed_idx(O("mlp_toa1.weight"), nh2, 2), nullptr, rc);
    d.toa1_b = add_gather(h, iota_off(O("mlp_toa1.bias"), nh2), nullptr, rc);
    if (lstm) {
        d.toa2_wt = add_gather(h, transposed_idx(O("mlp_toa2.weight"), nh2, 2), nullptr, rc);
        d.toa2_b = add_gather(h, iota_off(O("mlp_toa2.bias"), nh2), nullptr, rc);
    }
    d.init_wt = add_gather(h, transposed_idx(O("mlp_initial.weight"), nh1, nxp), nullptr, rc);
    d.init_b = add_gather(h, iota_off(O("mlp_initial.bias"), nh1), nullptr, rc);
    d.s1_wt = add_gather(h, transposed_idx(O("mlp_surface1.weight"), nh1, nxs), nullptr, rc);
    d.s1_b = add_gather(h, iota_off(O("mlp_surface1.bias"), nh1), nullptr, rc);
    if (lstm) {
        d.s2_wt = add_gather(h, transposed_idx(O("mlp_surface2.weight"), nh1, nxs), nullptr, rc);
        d.s2_b = add_gather(h, iota_off(O("mlp_surface2.bias"), nh1), nullptr, rc);
    }
    std::vector<int> rowmap1, rowmap2;   // permuted row n' -> PyTorch gate row, per LSTM
    auto lstm_pack = [&](int nh, int K, int o_wih, int o_whh, int o_bih, int o_bhh, const float *&wih, const float *&bias,
                         const float *&whhp, float *&whhTp, float *&wihT, std::vector<int> &rowmap, const float *&whhm) {
        std::vector<float> w, b, bhn;
        std::vector<float> iw = index_values((size_t)4 * nh * K), ib = index_values((size_t)4 * nh);
        pack_ih(1, nh, K, iw.data(), ib.data(), ib.data(), w, b, bhn);      // b = 2*index (b_ih+b_hh of equal indices)
        wih = add_gather(h, to_int(w, o_wih), nullptr, rc);
        std::vector<int> b1(4 * nh), b2(4 * nh);
        rowmap.resize(4 * nh);
        for (int n = 0; n < 4 * nh; ++n) { rowmap[n] = (int)(b[n] * 0.5f); b1[n] = o_bih + rowmap[n]; b2[n] = o_bhh + rowmap[n]; }
        bias = add_gather(h, b1, &b2, rc);
        std::vector<float> ih = index_values((size_t)4 * nh * nh), pk(rec_packed_floats(1, nh));
        rec_pack_weights(1, nh, ih.data(), pk.data());
        whhp = add_gather(h, to_int(pk, o_whh), nullptr, rc);
        whhm = nullptr;
        if (nh % 16 == 0) {      // matrix-pipe training forward (lstm_rec4m_kernel<NH, true, true>, from 544 columns per call)
            std::vector<float> pm((size_t)4 * nh * nh);
            rec4m_pack_weights(nh, ih.data(), pm.data());
            whhm = add_gather(h, to_int(pm, o_whh), nullptr, rc);
        }
        std::vector<float> pkT(bwd_rec_packed_floats(nh));
        bwd_rec_pack_weights(nh, ih.data(), pkT.data());
        whhTp = (float *)add_gather(h, to_int(pkT, o_whh), nullptr, rc);
        // W_ih^T in permuted-column order: (K rows, 4nh cols): [k][n'] = W_ih[rowmap[n']][k]
        std::vector<int> t((size_t)K * 4 * nh);
        for (int k = 0; k < K; ++k) for (int n = 0; n < 4 * nh; ++n) t[(size_t)k * 4 * nh + n] = o_wih + rowmap[n] * K + k;
        wihT = (float *)add_gather(h, t, nullptr, rc);
    };
    // GRU (training keeps four per unit, n' = u*4 + [r, z, n, pad]; inference packs three, pack.h): explicit index maps; -1 = padding (gathers read 0, scatters skip).
    // The saved-gate buffer after BPTT holds [dr~, dz~, dn~, g_hn] per unit (train_rec.hip): W_ih / b_ih gradients take
    // columns 0,1,2, W_hh / b_hh gradients take columns 0,1 and 3 (as gate row n).
    struct GruMaps { std::vector<int> wih, whh, ba, bb; } gm1, gm2;
    auto gru_pack = [&](int nh, int K, int o_wih, int o_whh, int o_bih, int o_bhh, const float *&wih, const float *&bias,
                        const float *&bhn, const float *&whhp, float *&whhTp, float *&wihT, GruMaps &gm, const float *&whhm) {
        std::vector<int> iw((size_t)4 * nh * K, -1), b1(4 * nh, -1), b2(4 * nh, -1), ibhn(nh), t((size_t)K * 4 * nh, -1);
        gm.wih.assign((size_t)4 * nh * K, -1); gm.whh.assign((size_t)4 * nh * nh, -1); gm.ba.assign(4 * nh, -1); gm.bb.assign(4 * nh, -1);
        for (int u = 0; u < nh; ++u) {
            for (int g = 0; g < 3; ++g) {
                const int src = g * nh + u, dst = u * 4 + g;
                for (int k = 0; k < K; ++k) {
                    iw[(size_t)dst * K + k] = o_wih + src * K + k;
                    t[(size_t)k * 4 * nh + dst] = o_wih + src * K + k;
                    gm.wih[(size_t)dst * K + k] = o_wih + src * K + k;
                }
                b1[dst] = o_bih + src;
                gm.ba[dst] = o_bih + src;
                if (g < 2) {
                    b2[dst] = o_bhh + src;
                    gm.bb[dst] = o_bhh + src;
                    for (int k = 0; k < nh; ++k) gm.whh[(size_t)dst * nh + k] = o_whh + src * nh + k;
                }
            }
            ibhn[u] = o_bhh + 2 * nh + u;
            gm.bb[u * 4 + 3] = o_bhh + 2 * nh + u;
            for (int k = 0; k < nh; ++k) gm.whh[(size_t)(u * 4 + 3) * nh + k] = o_whh + (2 * nh + u) * nh + k;
        }
        wih = add_gather(h, iw, nullptr, rc);
        bias = add_gather(h, b1, &b2, rc);
        bhn = add_gather(h, ibhn, nullptr, rc);
        std::vector<float> ih = index_values((size_t)3 * nh * nh), pk(rec_packed_floats(0, nh));
        gru2_pack_weights(nh, ih.data(), pk.data());            // the training forward is gru_rec2_kernel<NH, true> (round 3)
        whhp = add_gather(h, to_int(pk, o_whh), nullptr, rc);
        std::vector<float> pkT(bwd_rec_packed_floats_gru(nh));
        bwd_rec_pack_weights_gru(nh, ih.data(), pkT.data());
        whhTp = (float *)add_gather(h, to_int(pkT, o_whh), nullptr, rc);
        wihT = (float *)add_gather(h, t, nullptr, rc);
        whhm = nullptr;
        if (nh <= 128 && nh % 16 == 0) {      // matrix-pipe kernel at shard size (gru_rec4m_kernel<NH, true, TRAIN>): zero fourth rows -> index -1
            std::vector<float> i1((size_t)3 * nh * nh), pm((size_t)4 * nh * nh);
            for (size_t i = 0; i < i1.size(); ++i) i1[i] = (float)(i + 1);
            gru4m_pack_weights(nh, i1.data(), pm.data());
            std::vector<int> im(pm.size());
            for (size_t i = 0; i < pm.size(); ++i) im[i] = pm[i] == 0.0f ? -1 : o_whh + (int)pm[i] - 1;
            whhm = add_gather(h, im, nullptr, rc);
        }
    };
    // the two deterministic layers in execution order: (rnn1, rnn2), or (rnn0, rnn1) for the stochastic variant
    const std::string la = st ? "rnn0" : "rnn1", lb = st ? "rnn1" : "rnn2";
    auto ON = [&](const std::string &n) { return O(n.c_str()); };
    if (lstm) {
        lstm_pack(nh1, nin1, ON(la + ".weight_ih_l0"), ON(la + ".weight_hh_l0"), ON(la + ".bias_ih_l0"), ON(la + ".bias_hh_l0"),
                  d.wih1, d.bias1, d.whh1p, h->whh1Tp, h->wih1T, rowmap1, d.whh1m);
        lstm_pack(nh2, nh1, ON(lb + ".weight_ih_l0"), ON(lb + ".weight_hh_l0"), ON(lb + ".bias_ih_l0"), ON(lb + ".bias_hh_l0"),
                  d.wih2, d.bias2, d.whh2p, h->whh2Tp, h->wih2T, rowmap2, d.whh2m);
    } else {
        gru_pack(nh1, nin1, O("rnn1.weight_ih_l0"), O("rnn1.weight_hh_l0"), O("rnn1.bias_ih_l0"), O("rnn1.bias_hh_l0"),
                 d.wih1, d.bias1, d.bhn1, d.whh1p, h->whh1Tp, h->wih1T, gm1, d.whh1m);
        gru_pack(nh2, nh1, O("rnn2.weight_ih_l0"), O("rnn2.weight_hh_l0"), O("rnn2.bias_ih_l0"), O("rnn2.bias_hh_l0"),
                 d.wih2, d.bias2, d.bhn2, d.whh2p, h->whh2Tp, h->wih2T, gm2, d.whh2m);
    }
    d.lat_wt = add_gather(h, transposed_idx(O("mlp_latent.weight"), nm, nh2), nullptr, rc);
    d.lat_b = add_gather(h, iota_off(O("mlp_latent.bias"), nm), nullptr, rc);
    d.out_w = add_gather(h, iota_off(O("mlp_output.weight"), c.ny * nm), nullptr, rc);
    d.out_b = add_gather(h, iota_off(O("mlp_output.bias"), c.ny), nullptr, rc);
    d.sfo_w = add_gather(h, iota_off(O("mlp_surface_output.weight"), c.ny_sfc * nh2), nullptr, rc);
    d.sfo_b = add_gather(h, iota_off(O("mlp_surface_output.bias"), c.ny_sfc), nullptr, rc);

    // ---- gradient scatter maps -------------------------------------------------------------------------
    auto wmap = [&](const std::vector<int> &rowmap, int K, int o) {
        std::vector<int> v((size_t)rowmap.size() * K);
        for (size_t n = 0; n < rowmap.size(); ++n) for (int k = 0; k < K; ++k) v[n * K + k] = o + rowmap[n] * K + k;
        return v;
    };
    auto bmap = [&](const std::vector<int> &rowmap, int o) { std::vector<int> v(rowmap.size()); for (size_t n = 0; n < rowmap.size(); ++n) v[n] = o + rowmap[n]; return v; };
    if (lstm) {
        h->map_wih1 = upload_idx(h, wmap(rowmap1, nin1, ON(la + ".weight_ih_l0")), rc);
        h->map_whh1 = upload_idx(h, wmap(rowmap1, nh1, ON(la + ".weight_hh_l0")), rc);
        h->map_b1a = upload_idx(h, bmap(rowmap1, ON(la + ".bias_ih_l0")), rc);
        h->map_b1b = upload_idx(h, bmap(rowmap1, ON(la + ".bias_hh_l0")), rc);
        h->map_wih2 = upload_idx(h, wmap(rowmap2, nh1, ON(lb + ".weight_ih_l0")), rc);
        h->map_whh2 = upload_idx(h, wmap(rowmap2, nh2, ON(lb + ".weight_hh_l0")), rc);
        h->map_b2a = upload_idx(h, bmap(rowmap2, ON(lb + ".bias_ih_l0")), rc);
        h->map_b2b = upload_idx(h, bmap(rowmap2, ON(lb + ".bias_hh_l0")), rc);
    } else {
        h->map_wih1 = upload_idx(h, gm1.wih, rc); h->map_whh1 = upload_idx(h, gm1.whh, rc);
        h->map_b1a = upload_idx(h, gm1.ba, rc); h->map_b1b = upload_idx(h, gm1.bb, rc);
        h->map_wih2 = upload_idx(h, gm2.wih, rc); h->map_whh2 = upload_idx(h, gm2.whh, rc);
        h->map_b2a = upload_idx(h, gm2.ba, rc); h->map_b2b = upload_idx(h, gm2.bb, rc);
    }
    {   // head partial layout: [W_out | b_out | W_lat | b_lat | W_sfo | b_sfo]
        std::vector<int> v;
        auto app = [&](const char *n, int cnt) { const int o = O(n); for (int i = 0; i < cnt; ++i) v.push_back(o + i); };
        app("mlp_output.weight", c.ny * nm); app("mlp_output.bias", c.ny);
        app("mlp_latent.weight", nm * nh2); app("mlp_latent.bias", nm);
        app("mlp_surface_output.weight", c.ny_sfc * nh2); app("mlp_surface_output.bias", c.ny_sfc);
        h->map_head = upload_idx(h, v, rc);
    }
    {   // prep partial layout: [W_init | b_init | W_s1 | b_s1 | W_s2 | b_s2 | W_toa1 | b_toa1 | W_toa2 | b_toa2]
        // (GRU: the cell-state MLPs do not exist; their slots are written as zeros by the kernel and dropped here)
        std::vector<int> v;
        auto app = [&](const char *n, int cnt) { const int o = O(n); for (int i = 0; i < cnt; ++i) v.push_back(o < 0 ? -1 : o + i); };
        app("mlp_initial.weight", nh1 * nxp); app("mlp_initial.bias", nh1);
        app("mlp_surface1.weight", nh1 * nxs); app("mlp_surface1.bias", nh1);
        app("mlp_surface2.weight", nh1 * nxs); app("mlp_surface2.bias", nh1);
        app("mlp_toa1.weight", nh2 * 2); app("mlp_toa1.bias", nh2);
        app("mlp_toa2.weight", nh2 * 2); app("mlp_toa2.bias", nh2);
        h->map_prep = upload_idx(h, v, rc);
    }

    // ---- activations per slot + work buffers ----------------------------------------------------------------
    const size_t Bm = max_batch, LB = (size_t)L * Bm;
    for (int t = 0; t < max_window && rc == CSA_OK; ++t) {
        Slot s;
        s.X16 = dalloc<float>(h, Bm * L * nxp, rc); s.xs = dalloc<float>(h, Bm * nxs, rc);
        s.X1 = dalloc<float>(h, LB * nin1, rc);
        s.GP1 = dalloc<float>(h, LB * 4 * nh1, rc); s.C1 = dalloc<float>(h, (LB + Bm) * nh1, rc);
        s.H1lev = dalloc<float>(h, LB * nh1, rc); s.H1seq = dalloc<float>(h, (LB + Bm) * nh1, rc);
        s.GP2 = dalloc<float>(h, LB * 4 * nh2, rc); s.C2 = dalloc<float>(h, (LB + Bm) * nh2, rc);
        s.H2 = dalloc<float>(h, (LB + Bm) * nh2, rc);
        s.Z = dalloc<float>(h, LB * nm, rc); s.hc0 = dalloc<float>(h, 4 * Bm * nhm, rc);
        h->slots.push_back(s);
    }
    if (st && rc == CSA_OK) {
        // the stochastic layer: ONE handle; its weight layouts become gathers of the flat parameters (so Adam's re-pack
        // refreshes them too), its saved activations are per-slot buffers swapped in before every call
        rc = csa_stoch_lstm4_create(nh1, nh2, p->rnn2_weight_encoder, (int)LB, &h->stoch);
        if (rc == CSA_OK) rc = csa_stoch_enable_training(h->stoch);
        if (rc == CSA_OK) {
            const int oe = O("rnn2.weight_encoder"), W5 = 5 * nh2;
            h->off_enc = oe;
            auto reg = [&](float *dst, const std::vector<int> &idx) {
                Gather g; g.n = (int)idx.size(); g.dst = dst; g.idx = upload_idx(h, idx, rc); g.idx2 = nullptr;
                h->gathers.push_back(g);
            };
            std::vector<float> iv = index_values((size_t)nh2 * W5), pk((size_t)nh2 * W5);
            stoch_pack_rows(nh2, 5, iv.data(), W5, 0, pk.data());                 // forward packing of the hidden half
            reg(h->stoch->wp_a, to_int(pk, oe + nh1 * W5));
            stoch_pack_t(nh2, W5, iv.data(), pk.data());                          // BPTT packing
            reg(h->stoch->wT_a, to_int(pk, oe + nh1 * W5));
            std::vector<int> t((size_t)W5 * nh1);                                 // (5H, nx) transposed input half for the NT GEMM
            for (int k = 0; k < nh1; ++k) for (int n = 0; n < W5; ++n) t[(size_t)n * nh1 + k] = oe + k * W5 + n;
            reg(h->stoch->w_in_t, t);
            reg(h->stoch->w_ref_in, iota_off(oe, nh1 * W5));
        }
        for (Slot &S : h->slots) {
            S.Hb = dalloc<float>(h, LB * nh2, rc); S.Zs = dalloc<float>(h, LB * nh2, rc);
            S.sXP = dalloc<float>(h, LB * 5 * nh2, rc); S.sH = dalloc<float>(h, (LB + Bm) * nh2, rc); S.sC = dalloc<float>(h, (LB + Bm) * nh2, rc);
        }
        h->dscr = dalloc<float>(h, 2 * Bm * nhm, rc);
    }
    h->dH2 = dalloc<float>(h, LB * nh2, rc); h->dH1 = dalloc<float>(h, LB * nh1, rc);
    h->dX1 = dalloc<float>(h, LB * nin1, rc);
    h->dhc1 = dalloc<float>(h, 2 * Bm * nhm, rc); h->dhc2 = dalloc<float>(h, 2 * Bm * nhm, rc);
    if (rc == CSA_OK && (hipMemset(h->dhc1, 0, sizeof(float) * 2 * Bm * nhm) != hipSuccess ||
                         hipMemset(h->dhc2, 0, sizeof(float) * 2 * Bm * nhm) != hipSuccess)) rc = CSA_ERR_HIP;
    h->nsplit = 192;
    size_t pf = (size_t)h->nsplit * 4 * nhm * (nin1 > nhm ? nin1 : nhm);
    const size_t pcol = Bm * (size_t)std::max(head_bwd_partial_floats(c), prep_bwd_partial_floats(c));
    h->part_floats = pf > pcol ? pf : pcol;
    h->part = dalloc<float>(h, h->part_floats, rc);
    h->part_b = dalloc<float>(h, (size_t)h->nsplit * 4 * std::max(c.nh1, c.nh2), rc);
    h->rtmp = dalloc<float>(h, (size_t)32 * std::max(head_bwd_partial_floats(c), prep_bwd_partial_floats(c)), rc);
    h->rtmp2 = dalloc<float>(h, (size_t)32 * prep_bwd_partial_floats(c), rc);
    h->part2 = dalloc<float>(h, Bm * (size_t)prep_bwd_partial_floats(c), rc);
    h->arena_floats = (size_t)h->nsplit * 4 * ((size_t)c.nh2 * c.nh1 + c.nh2 + (size_t)c.nh2 * c.nh2 + (size_t)c.nh1 * nin1 + c.nh1 + (size_t)c.nh1 * c.nh1);
    h->arena = dalloc<float>(h, h->arena_floats, rc);
    h->samp = dalloc<float>(h, (size_t)max_window * Bm * 9, rc);
    h->ecoef = dalloc<float>(h, Bm, rc);
    h->sp = dalloc<float>(h, (size_t)max_window * Bm, rc);
    if (rc == CSA_OK) rc = repack(h, 0);
    if (rc == CSA_OK && hipDeviceSynchronize() != hipSuccess) rc = CSA_ERR_HIP;
    if (rc != CSA_OK) {
        if (h->stoch) (void)csa_stoch_destroy(h->stoch);
        for (void *q : h->owned) (void)hipFree(q);
        delete h;
        return rc;
    }
    *out = h;
    return CSA_OK;
}

extern "C" int csa_train_destroy(csa_trainer *h)
{
    if (!h) return CSA_ERR_ARG;
    prof_collect(h);
    if (h->stoch) (void)csa_stoch_destroy(h->stoch);
    for (hipEvent_t e : h->ev_free) (void)hipEventDestroy(e);
    for (void *q : h->owned) (void)hipFree(q);
    delete h;
    return CSA_OK;
}
extern "C" int csa_train_num_params(const csa_trainer *h) { return h ? h->nparam : CSA_ERR_ARG; }
extern "C" int csa_train_num_tensors(const csa_trainer *h) { return h ? (int)h->info.size() : CSA_ERR_ARG; }
extern "C" int csa_train_param_info(const csa_trainer *h, int i, const char **name, int *offset, int *rows, int *cols)
{
    if (!h || i < 0 || i >= (int)h->info.size()) return CSA_ERR_ARG;
    if (name) *name = h->info[i].name.c_str();
    if (offset) *offset = h->info[i].off;
    if (rows) *rows = h->info[i].rows;
    if (cols) *cols = h->info[i].cols;
    return CSA_OK;
}
extern "C" float *csa_train_params(csa_trainer *h) { return h ? h->params : nullptr; }
// which = 0 parameters, 1 Adam first moment, 2 Adam second moment; dir = 0 copy out to buf, 1 copy in from buf
// (device buffers of csa_train_num_params floats).  Copying parameters in re-packs the kernel layouts.
extern "C" int csa_train_copy_state(csa_trainer *h, int which, int dir, float *buf, void *stream)
{
    if (!h || !buf || which < 0 || which > 2 || dir < 0 || dir > 1) { csa_set_error_msg("csa_train_copy_state: bad argument"); return CSA_ERR_ARG; }
    float *st = which == 0 ? h->params : (which == 1 ? h->adam_m : h->adam_v);
    hipStream_t s = (hipStream_t)stream;
    CSA_HIP_CHECK(hipMemcpyAsync(dir ? st : buf, dir ? buf : st, sizeof(float) * h->nparam, hipMemcpyDeviceToDevice, s));
    if (dir == 1 && which == 0) return repack(h, s);
    return CSA_OK;
}
extern "C" int csa_train_sync_params(csa_trainer *h, void *stream) { return h ? repack(h, (hipStream_t)stream) : CSA_ERR_ARG; }

// training forward of one LSTM: two columns per workgroup (packed FMA), or four on the matrix pipe from CSA_REC4_MIN_BATCH columns
// per call (the kernel classes differ in the order of the k-sum only; what they save for BPTT has the same layout)
static int rec_train_lstm(int nh, const float *whhp, const float *whhm, float *P, const float *h0, const float *c0, float *Hout, int B, int L,
                          int reverse_out, float *Hseq, float *Cseq, hipStream_t s)
{
    if (whhm && rec4m_selected(1, nh, B)) return launch_rec4m_train(nh, whhm, P, h0, c0, Hout, B, L, reverse_out, Hseq, Cseq, s);
    return launch_rec_train(nh, whhp, P, h0, c0, Hout, B, L, reverse_out, Hseq, Cseq, s);
}

extern "C" int csa_train_forward(csa_trainer *h, int slot, int B, const float *x_main_n, const float *x_sfc_n,
                                 const float *mem_in, float *out, float *out_sfc, float *mem_out, void *stream)
{
    if (!h || slot < 0 || slot >= h->max_window || B <= 0 || B > h->max_batch || !x_main_n || !x_sfc_n || !mem_in || !out || !out_sfc || !mem_out) {
        csa_set_error_msg("csa_train_forward: bad argument");
        return CSA_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const csa_config &c = h->dm.cfg;
    if (c.add_stochastic_layer) { csa_set_error_msg("csa_train_forward: the stochastic variant takes its noise explicitly (csa_train_forward_noise)"); return CSA_ERR_ARG; }
    const int L = c.nlev, nh1 = c.nh1, nh2 = c.nh2, nm = c.nh_mem, nhm = nh1 > nh2 ? nh1 : nh2;
    Slot &S = h->slots[slot];
    int rc;
    if ((rc = launch_prep_train(h->dm, B, 1, x_main_n, x_sfc_n, mem_in, S.X1, S.hc0, S.X16, S.xs, s))) return rc;
    {
        StageTimer tm(h, 2, s);
        if ((rc = launch_proj_gemm(S.X1, h->dm.wih1, h->dm.bias1, S.GP1, L * B, 4 * nh1, nh1 + nm, s))) return rc;
    }
    {
        StageTimer tm(h, 0, s);
        if (c.use_lstm) rc = rec_train_lstm(nh1, h->dm.whh1p, h->dm.whh1m, S.GP1, S.hc0, S.hc0 + (size_t)B * nhm, S.H1lev, B, L, 1, S.H1seq, S.C1, s);
        else rc = h->dm.whh1m && gru4m_selected(nh1, B) ? launch_rec4m_train_gru(nh1, h->dm.whh1m, h->dm.bhn1, S.GP1, S.hc0, S.H1lev, B, L, 1, S.H1seq, s)
                                                        : launch_rec_train_gru(nh1, h->dm.whh1p, h->dm.bhn1, S.GP1, S.hc0, S.H1lev, B, L, 1, S.H1seq, s);
        if (rc) return rc;
    }
    {
        StageTimer tm(h, 2, s);
        if ((rc = launch_proj_gemm(S.H1lev, h->dm.wih2, h->dm.bias2, S.GP2, L * B, 4 * nh2, nh1, s))) return rc;
    }
    {   // rnn2: level order == sequence order, so the hidden sequence itself carries the extra slot 0
        StageTimer tm(h, 0, s);
        if (c.use_lstm) rc = rec_train_lstm(nh2, h->dm.whh2p, h->dm.whh2m, S.GP2, S.hc0 + (size_t)2 * B * nhm, S.hc0 + (size_t)3 * B * nhm,
                                            S.H2 + (size_t)B * nh2, B, L, 0, S.H2, S.C2, s);
        else rc = h->dm.whh2m && gru4m_selected(nh2, B)
                      ? launch_rec4m_train_gru(nh2, h->dm.whh2m, h->dm.bhn2, S.GP2, S.hc0 + (size_t)2 * B * nhm, S.H2 + (size_t)B * nh2, B, L, 0, S.H2, s)
                      : launch_rec_train_gru(nh2, h->dm.whh2p, h->dm.bhn2, S.GP2, S.hc0 + (size_t)2 * B * nhm, S.H2 + (size_t)B * nh2, B, L, 0, S.H2, s);
        if (rc) return rc;
    }
    if ((rc = launch_head(h->dm, B, HEAD_RAW, S.H2 + (size_t)B * nh2, x_main_n, nullptr, out, out_sfc, S.Z, s))) return rc;
    CSA_HIP_CHECK(hipMemcpyAsync(mem_out, S.Z, sizeof(float) * (size_t)L * B * nm, hipMemcpyDeviceToDevice, s));
    return CSA_OK;
}

// ---- the stochastic variant (models.py:464-474,521-534): rnn0 down (noise init) -> rnn1 up (surface init) -> stochastic LSTM down
// (TOA init).  Layer "1" of the trainer is rnn0 (X1 in LEVEL order), layer "2" is rnn1 (its input H1lev = rnn0's sequence flipped).
extern "C" int csa_train_forward_noise(csa_trainer *h, int slot, int B, const float *x_main_n, const float *x_sfc_n, const float *mem_in,
                                       const float *hx0, const float *cx0, const float *eps, float *out, float *out_sfc,
                                       float *mem_out, void *stream)
{
    if (!h || !h->stoch || slot < 0 || slot >= h->max_window || B <= 0 || B > h->max_batch || !x_main_n || !x_sfc_n || !mem_in || !hx0 ||
        !cx0 || !eps || !out || !out_sfc || !mem_out) {
        csa_set_error_msg("csa_train_forward_noise: bad argument (or the trainer was not created with add_stochastic_layer)");
        return CSA_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const csa_config &c = h->dm.cfg;
    const int L = c.nlev, nh = c.nh1, nm = c.nh_mem;
    Slot &S = h->slots[slot];
    int rc;
    if ((rc = launch_prep_train(h->dm, B, 1, x_main_n, x_sfc_n, mem_in, S.X1, S.hc0, S.X16, S.xs, s))) return rc;
    if ((rc = launch_proj_gemm(S.X1, h->dm.wih1, h->dm.bias1, S.GP1, L * B, 4 * nh, nh + nm, s))) return rc;
    // rnn0 runs downward over the level-ordered rows; its output is stored flipped = the sequence order of the upward rnn1
    if ((rc = rec_train_lstm(nh, h->dm.whh1p, h->dm.whh1m, S.GP1, hx0, cx0, S.H1lev, B, L, 1, S.H1seq, S.C1, s))) return rc;
    if ((rc = launch_proj_gemm(S.H1lev, h->dm.wih2, h->dm.bias2, S.GP2, L * B, 4 * nh, nh, s))) return rc;
    if ((rc = rec_train_lstm(nh, h->dm.whh2p, h->dm.whh2m, S.GP2, S.hc0, S.hc0 + (size_t)B * nh, S.Hb, B, L, 1, S.H2, S.C2, s))) return rc;
    csa_stoch *st = h->stoch;
    st->XP = S.sXP; st->Hseq = S.sH; st->Cseq = S.sC;
    if ((rc = csa_stoch_lstm4_forward_train(st, L, B, S.Hb, S.hc0 + (size_t)2 * B * nh, S.hc0 + (size_t)3 * B * nh, eps, S.Zs, nullptr,
                                            nullptr, s))) return rc;
    S.eps = eps;
    if ((rc = launch_head(h->dm, B, HEAD_RAW, S.Zs, x_main_n, nullptr, out, out_sfc, S.Z, s))) return rc;
    CSA_HIP_CHECK(hipMemcpyAsync(mem_out, S.Z, sizeof(float) * (size_t)L * B * nm, hipMemcpyDeviceToDevice, s));
    return CSA_OK;
}

static int train_backward_stoch(csa_trainer *h, int slot, int B, const float *d_out, const float *d_out_sfc, const float *d_mem_out,
                                float *d_mem_in, float *grads, hipStream_t s)
{
    const csa_config &c = h->dm.cfg;
    const int L = c.nlev, nh = c.nh1, nm = c.nh_mem, nin1 = nh + nm, M = L * B, ns = h->nsplit;
    Slot &S = h->slots[slot];
    if (!S.eps) { csa_set_error_msg("csa_train_backward: no forward_noise call recorded for this slot"); return CSA_ERR_ARG; }
    int rc;
    if ((rc = launch_head_bwd(h->dm, B, d_out, d_out_sfc, d_mem_out, S.Z, S.Zs, h->dH2, h->part, s))) return rc;
    if ((rc = launch_reduce_partials_2stage(h->part, B, head_bwd_partial_floats(c), h->map_head, nullptr, grads, h->rtmp, 32, s))) return rc;
    // stochastic LSTM: BPTT + its input / weight gradients; d(h0, c0) = the gradients of the TOA MLPs' outputs
    csa_stoch *st = h->stoch;
    st->XP = S.sXP; st->Hseq = S.sH; st->Cseq = S.sC;
    if ((rc = csa_stoch_lstm4_backward(st, L, B, S.Hb, S.eps, h->dH2, nullptr, nullptr, h->dH1, h->dhc2, h->dhc2 + (size_t)B * nh, nullptr,
                                       grads + h->off_enc, s))) return rc;
    // rnn1 (upward; dH1 is in level order, the recurrence ran in sequence order); d(h0, c0) -> the surface MLPs
    if ((rc = launch_bwd_rec(nh, h->whh2Tp, S.GP2, S.C2, h->dH1, h->dhc1, h->dhc1 + (size_t)B * nh, B, L, 1, s))) return rc;
    if ((rc = launch_proj_gemm(S.GP2, h->wih2T, nullptr, h->dH2, M, nh, 4 * nh, s))) return rc;          // d(H1lev), rnn1's sequence order
    if ((rc = launch_gemm_tn_partial(S.GP2, 4 * nh, S.H1lev, nh, h->part, M, 4 * nh, nh, ns, s))) return rc;
    if ((rc = launch_reduce_partials(h->part, ns, 4 * nh * nh, h->map_wih2, nullptr, grads, s))) return rc;
    if ((rc = launch_gemm_tn_partial(S.GP2, 4 * nh, S.H2, nh, h->part, M, 4 * nh, nh, ns, s))) return rc;
    if ((rc = launch_reduce_partials(h->part, ns, 4 * nh * nh, h->map_whh2, nullptr, grads, s))) return rc;
    if ((rc = launch_colsum_partial(S.GP2, h->part, M, 4 * nh, ns, s))) return rc;
    if ((rc = launch_reduce_partials(h->part, ns, 4 * nh, h->map_b2a, h->map_b2b, grads, s))) return rc;
    // rnn0 (downward; its output row of step t sits at flipped position L-1-t): initial state was noise, its gradient is dropped
    if ((rc = launch_bwd_rec(nh, h->whh1Tp, S.GP1, S.C1, h->dH2, h->dscr, h->dscr + (size_t)B * nh, B, L, 1, s))) return rc;
    if ((rc = launch_proj_gemm(S.GP1, h->wih1T, nullptr, h->dX1, M, nin1, 4 * nh, s))) return rc;
    if ((rc = launch_gemm_tn_partial(S.GP1, 4 * nh, S.X1, nin1, h->part, M, 4 * nh, nin1, ns, s))) return rc;
    if ((rc = launch_reduce_partials(h->part, ns, 4 * nh * nin1, h->map_wih1, nullptr, grads, s))) return rc;
    if ((rc = launch_gemm_tn_partial(S.GP1, 4 * nh, S.H1seq, nh, h->part, M, 4 * nh, nh, ns, s))) return rc;
    if ((rc = launch_reduce_partials(h->part, ns, 4 * nh * nh, h->map_whh1, nullptr, grads, s))) return rc;
    if ((rc = launch_colsum_partial(S.GP1, h->part, M, 4 * nh, ns, s))) return rc;
    if ((rc = launch_reduce_partials(h->part, ns, 4 * nh, h->map_b1a, h->map_b1b, grads, s))) return rc;
    if ((rc = launch_prep_bwd(h->dm, B, h->dX1, S.X1, S.X16, S.xs, S.hc0, h->dhc1, h->dhc2, d_mem_in, h->part, s))) return rc;
    return launch_reduce_partials_2stage(h->part, B, prep_bwd_partial_floats(c), h->map_prep, nullptr, grads, h->rtmp, 32, s);
}

extern "C" int csa_train_backward(csa_trainer *h, int slot, int B, const float *d_out, const float *d_out_sfc,
                                  const float *d_mem_out, float *d_mem_in, float *grads, void *stream)
{
    if (!h || slot < 0 || slot >= h->max_window || B <= 0 || B > h->max_batch || !d_out || !d_out_sfc || !grads) {
        csa_set_error_msg("csa_train_backward: bad argument");
        return CSA_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const csa_config &c = h->dm.cfg;
    if (c.add_stochastic_layer) return train_backward_stoch(h, slot, B, d_out, d_out_sfc, d_mem_out, d_mem_in, grads, s);
    const int L = c.nlev, nh1 = c.nh1, nh2 = c.nh2, nm = c.nh_mem, nin1 = nh1 + nm, M = L * B, ns = h->nsplit;
    const int nhm = nh1 > nh2 ? nh1 : nh2;
    Slot &S = h->slots[slot];
    int rc;
    // head
    if ((rc = launch_head_bwd(h->dm, B, d_out, d_out_sfc, d_mem_out, S.Z, S.H2 + (size_t)B * nh2, h->dH2, h->part, s))) return rc;
    // every reduction of this call is queued (distinct targets, a partial buffer each) and runs as one launch at its end
    ReduceJobs rq;
    if ((rc = reduce_queue_add_2stage(rq, h->part, B, head_bwd_partial_floats(c), h->map_head, nullptr, h->rtmp, 32, s))) return rc;
    float *pa = h->arena;
    auto wgrad = [&](const float *A, int lda, const float *Bm, int ldb, int N1, int N2, const int *map, const int *ba, const int *bb) -> int {
        float *cp = pa, *cb = nullptr;
        pa += (size_t)ns * N1 * N2;
        if (ba) { cb = pa; pa += (size_t)ns * N1; }
        int r = ba ? launch_gemm_tn_partial_cs(A, lda, Bm, ldb, cp, cb, M, N1, N2, ns, s) : launch_gemm_tn_partial(A, lda, Bm, ldb, cp, M, N1, N2, ns, s);
        if (r) return r;
        if ((r = reduce_queue_add(rq, cp, ns, N1 * N2, map, nullptr))) return r;
        return ba ? reduce_queue_add(rq, cb, ns, N1, ba, bb) : CSA_OK;
    };
    // rnn2 (downward): BPTT, then input / weight gradients from dP2 (stored in place in GP2)
    { StageTimer tm(h, 1, s);
    if (c.use_lstm) {
        if ((rc = launch_bwd_rec(nh2, h->whh2Tp, S.GP2, S.C2, h->dH2, h->dhc2, h->dhc2 + (size_t)B * nhm, B, L, 0, s))) return rc;
    } else {
        if ((rc = launch_bwd_rec_gru(nh2, h->whh2Tp, S.GP2, S.H2, h->dH2, h->dhc2, B, L, 0, s))) return rc;
    } }
    { StageTimer tm(h, 3, s);
    if ((rc = launch_proj_gemm(S.GP2, h->wih2T, nullptr, h->dH1, M, nh1, 4 * nh2, s))) return rc; }
    if (!h->defer) {
        if ((rc = wgrad(S.GP2, 4 * nh2, S.H1lev, nh1, 4 * nh2, nh1, h->map_wih2, h->map_b2a, h->map_b2b))) return rc;
        if ((rc = wgrad(S.GP2, 4 * nh2, S.H2, nh2, 4 * nh2, nh2, h->map_whh2, nullptr, nullptr))) return rc;
    }
    // rnn1 (upward): dH1 is in level order, the recurrence runs in sequence order
    { StageTimer tm(h, 1, s);
    if (c.use_lstm) {
        if ((rc = launch_bwd_rec(nh1, h->whh1Tp, S.GP1, S.C1, h->dH1, h->dhc1, h->dhc1 + (size_t)B * nhm, B, L, 1, s))) return rc;
    } else {
        if ((rc = launch_bwd_rec_gru(nh1, h->whh1Tp, S.GP1, S.H1seq, h->dH1, h->dhc1, B, L, 1, s))) return rc;
    } }
    { StageTimer tm(h, 3, s);
    if ((rc = launch_proj_gemm(S.GP1, h->wih1T, nullptr, h->dX1, M, nin1, 4 * nh1, s))) return rc; }
    if (!h->defer) {
        if ((rc = wgrad(S.GP1, 4 * nh1, S.X1, nin1, 4 * nh1, nin1, h->map_wih1, h->map_b1a, h->map_b1b))) return rc;
        if ((rc = wgrad(S.GP1, 4 * nh1, S.H1seq, nh1, 4 * nh1, nh1, h->map_whh1, nullptr, nullptr))) return rc;
    } else {
        if (!h->pending.empty() && h->pending_B != B) { csa_set_error_msg("csa_train_backward(deferred): batch size changed inside a window"); return CSA_ERR_ARG; }
        if ((int)h->pending.size() >= TN_MAX_SEGS) { csa_set_error_msg("csa_train_backward(deferred): flush before more than 8 pending steps"); return CSA_ERR_ARG; }
        h->pending.push_back(slot);
        h->pending_B = B;
    }
    // mlp_initial / surface / TOA MLPs, gradient w.r.t. the incoming memory
    if ((rc = launch_prep_bwd(h->dm, B, h->dX1, S.X1, S.X16, S.xs, S.hc0, h->dhc1, h->dhc2, d_mem_in, h->part2, s))) return rc;
    if ((rc = reduce_queue_add_2stage(rq, h->part2, B, prep_bwd_partial_floats(c), h->map_prep, nullptr, h->rtmp2, 32, s))) return rc;
    return launch_reduce_queue(rq, grads, s);
}

extern "C" int csa_train_set_deferred(csa_trainer *h, int enable)
{
    if (!h) return CSA_ERR_ARG;
    if (!h->pending.empty()) { csa_set_error_msg("csa_train_set_deferred: flush pending gradients first"); return CSA_ERR_ARG; }
    if (enable && h->stoch) { csa_set_error_msg("csa_train_set_deferred: not available for the stochastic variant"); return CSA_ERR_UNSUPPORTED; }
    h->defer = enable != 0;
    return CSA_OK;
}

// the W_ih / W_hh gradients of every backward call since the last flush, each as ONE split-M GEMM over all pending steps
extern "C" int csa_train_flush_wgrad(csa_trainer *h, float *grads, void *stream)
{
    if (!h || !grads) return CSA_ERR_ARG;
    if (h->pending.empty()) return CSA_OK;
    hipStream_t s = (hipStream_t)stream;
    const csa_config &c = h->dm.cfg;
    const int nh1 = c.nh1, nh2 = c.nh2, nin1 = nh1 + c.nh_mem, M = c.nlev * h->pending_B;
    const int nseg = (int)h->pending.size();
    const int ns = (h->nsplit / nseg) * nseg;      // same number of partials as one per-step call, shared by the segments
    int rc;
    StageTimer tm(h, 4, s);
    // the four GEMMs keep their partials in regions of the arena; ONE queued reduction at the end
    ReduceJobs rq;
    float *pa = h->arena;
    // ba / bb: the bias gradient (column sums of dP over all pending steps) comes out of the W_ih GEMM of each RNN
    auto run = [&](float *Slot::*a, float *Slot::*b, int N1, int N2, const int *map, const int *ba, const int *bb) {
        TnSegs g{};
        g.n = nseg;
        for (int i = 0; i < nseg; ++i) { g.A[i] = h->slots[h->pending[i]].*a; g.B[i] = h->slots[h->pending[i]].*b; }
        float *cp = pa, *cb = nullptr;
        pa += (size_t)ns * N1 * N2;
        if (ba) { cb = pa; pa += (size_t)ns * N1; }
        if ((rc = launch_gemm_tn_segs(g, N1, N2, cp, M, N1, N2, ns, 0, 0, s, cb))) return rc;
        if ((rc = reduce_queue_add(rq, cp, ns, N1 * N2, map, nullptr))) return rc;
        return ba ? reduce_queue_add(rq, cb, ns, N1, ba, bb) : CSA_OK;
    };
    if ((rc = run(&Slot::GP2, &Slot::H1lev, 4 * nh2, nh1, h->map_wih2, h->map_b2a, h->map_b2b))) return rc;
    if ((rc = run(&Slot::GP2, &Slot::H2, 4 * nh2, nh2, h->map_whh2, nullptr, nullptr))) return rc;
    if ((rc = run(&Slot::GP1, &Slot::X1, 4 * nh1, nin1, h->map_wih1, h->map_b1a, h->map_b1b))) return rc;
    if ((rc = run(&Slot::GP1, &Slot::H1seq, 4 * nh1, nh1, h->map_whh1, nullptr, nullptr))) return rc;
    if ((rc = launch_reduce_queue(rq, grads, s))) return rc;
    h->pending.clear();
    return CSA_OK;
}

__global__ void sp_kernel(const float *__restrict__ xs, int nxs, float a, float bconst, float *__restrict__ sp, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sp[i] = xs[(size_t)i * nxs] * a + bconst;
}

extern "C" int csa_train_loss(csa_trainer *h, int B, int Tw, float w_h, float w_w,
                              const float *pred, const float *pred_sfc, const float *tgt, const float *tgt_sfc,
                              const float *yto, const float *yto_sfc, const float *x_raw, const float *x_sfc_n,
                              float *scalars, float *d_pred, float *d_pred_sfc, void *stream)
{
    if (!h || B <= 0 || B > h->max_batch || Tw <= 0 || Tw > h->max_window || !pred || !pred_sfc || !tgt || !tgt_sfc || !yto || !yto_sfc || !x_raw || !x_sfc_n || !scalars) {
        csa_set_error_msg("csa_train_loss: bad argument");
        return CSA_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const int N = B * Tw;
    // surface pressure de-normalised from the normalised x_sfc (rnn/utils.py:1248); the two scalars were kept at create
    const float a = h->sp_scale, bc = h->sp_shift;
    hipLaunchKernelGGL(sp_kernel, dim3((N + 255) / 256), dim3(256), 0, s, x_sfc_n, h->dm.cfg.nx_sfc, a, bc, h->sp, N);
    return launch_loss(h->dm, h->hyai, h->hybi, B, Tw, w_h, w_w, pred, pred_sfc, tgt, tgt_sfc, yto, yto_sfc, x_raw, h->sp,
                       h->samp, h->ecoef, scalars, d_pred, d_pred_sfc, s);
}

extern "C" int csa_train_adam(csa_trainer *h, const float *grads, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, void *stream)
{
    if (!h || !grads || step <= 0) { csa_set_error_msg("csa_train_adam: bad argument"); return CSA_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_adam(h->params, grads, h->adam_m, h->adam_v, h->nparam, lr, beta1, beta2, eps, step, weight_decay, s);
    if (rc) return rc;
    return repack(h, s);
}

// ---- per-stage timing (HIP events on the caller's stream; measurement mode only) ----------------------------------------
static const char *const kTrainStages[CSA_TRAIN_NSTAGE] = {"fwd_rec", "bwd_rec", "fwd_proj_gemm", "dx_gemm", "wgrad_flush"};
extern "C" const char *csa_train_stage_name(int i) { return i >= 0 && i < CSA_TRAIN_NSTAGE ? kTrainStages[i] : ""; }
extern "C" int csa_train_set_profiling(csa_trainer *h, int enable)
{
    if (!h) return CSA_ERR_ARG;
    if (!enable) prof_collect(h);
    h->profiling = enable != 0;
    return CSA_OK;
}
extern "C" int csa_train_reset_profile(csa_trainer *h)
{
    if (!h) return CSA_ERR_ARG;
    prof_collect(h);
    for (int i = 0; i < CSA_TRAIN_NSTAGE; ++i) { h->acc_ms[i] = 0.0; h->acc_n[i] = 0; }
    return CSA_OK;
}
extern "C" int csa_train_get_profile(csa_trainer *h, double *avg_ms, long *launches, int n)
{
    if (!h || !avg_ms || n < CSA_TRAIN_NSTAGE) return CSA_ERR_ARG;
    prof_collect(h);
    for (int i = 0; i < CSA_TRAIN_NSTAGE; ++i) {
        avg_ms[i] = h->acc_n[i] ? h->acc_ms[i] / (double)h->acc_n[i] : 0.0;
        if (launches) launches[i] = h->acc_n[i];
    }
    return CSA_OK;
}
