// api.hip -- C ABI of libclimsim_amd.so (include/climsim_amd.h): handle management, host-side
// weight packing, and the launch sequence of one emulator call:
//
//   prep  ->  proj GEMM (W_ih1)  ->  rec (rnn1, upward)  ->  proj GEMM (W_ih2)  ->  rec (rnn2, downward)  ->  head
//
// All launches go to the caller's stream; nothing is allocated or synchronised inside a call.
#include "common.h"
#include "pack.h"
#include <mutex>
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>

static thread_local std::string g_err;
void csa_set_error(const char *what, hipError_t e)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
}
void csa_set_error_msg(const char *msg) { g_err = msg; }
extern "C" const char *csa_last_error(void) { return g_err.c_str(); }
extern "C" const char *csa_version(void) { return "climsim_amd 0.1 (gfx950)"; }

struct csa_emulator {
    DevModel dm;
    int max_batch;
    std::vector<void *> owned;   // every device allocation: parameters first, then scratch
    size_t n_param_allocs = 0;
    // scratch
    float *X1, *P, *H1, *H2, *hc0;
    csa_stoch *stoch = nullptr;  // add_stochastic_layer: the MyStochasticLSTMLayer4 stage (stoch.hip)
    float *ar_eps = nullptr, *ar_o6 = nullptr, *ar_sfc = nullptr, *ar_mem = nullptr, *ar_memT = nullptr;   // csa_forward_packed_noise scratch
    // optional per-kernel timing (csa_set_profiling): events bracket the 6 launches of a call
    bool profiling = false;
    hipEvent_t ev[7] = {};
    double acc_ms[6] = {};
    long n_prof = 0;
    bool pending = false;
    // column halves on two streams (run_forward_halves): side stream + fork / join events
    hipStream_t side = nullptr;
    hipEvent_t ov_ev[2] = {};
    int rec1_max_batch = 256;    // largest batch that uses the one-column-per-workgroup recurrent kernel (csa_set_rec1_max_batch)
    int halves = 2;              // two column halves on two streams (run_forward_halves): 0 off, 1 on, 2 auto (B >= 640)
};

namespace {

struct Uploader {
    csa_emulator *h;
    int rc = CSA_OK;
    float *alloc(size_t n)
    {
        void *p = nullptr;
        if (hipMalloc(&p, sizeof(float) * (n ? n : 1)) != hipSuccess) { rc = CSA_ERR_NOMEM; return nullptr; }
        h->owned.push_back(p);
        return (float *)p;
    }
    const float *up(const float *host, size_t n)
    {
        float *d = alloc(n);
        if (d && host && hipMemcpy(d, host, sizeof(float) * n, hipMemcpyHostToDevice) != hipSuccess) rc = CSA_ERR_HIP;
        return d;
    }
    const float *up(const std::vector<float> &v) { return up(v.data(), v.size()); }
};

int check_cfg(const csa_config &c)
{
    if (c.nlev <= 0 || c.nx <= 0 || c.nx_sfc < 7 || c.ny <= 0 || c.ny_sfc <= 0 || c.nh1 <= 0 || c.nh2 <= 0 || c.nh_mem < 0) {
        csa_set_error_msg("csa_create: bad sizes");
        return CSA_ERR_ARG;
    }
    if (c.mp_mode != 0 && c.mp_mode != 1 && c.mp_mode != -1 && c.mp_mode != -2) {
        csa_set_error_msg("csa_create: mp_mode must be 0, 1, -1 or -2 (models.py:203-227)");
        return CSA_ERR_ARG;
    }
    if (c.mp_mode < 0 && (c.ny != 6 || c.legacy)) {
        csa_set_error_msg("csa_create: mp_mode -1/-2 are current-generation models with ny == 6");
        return CSA_ERR_ARG;
    }
    if (c.mp_mode == 1 && c.ny != 5) {
        csa_set_error_msg("csa_create: mp_mode 1 requires ny == 5 (models.py:216-217)");
        return CSA_ERR_ARG;
    }
    if (c.add_stochastic_layer && (c.legacy || !c.use_lstm || c.nh1 != c.nh2 || c.nh_mem <= 0)) {
        // models.py:405-412 with use_lstm; the GRU flavour raises AttributeError upstream (mlp_toa2 missing)
        csa_set_error_msg("csa_create: add_stochastic_layer needs the current-generation LSTM with memory and nh1 == nh2");
        return CSA_ERR_UNSUPPORTED;
    }
    return CSA_OK;
}

int upload_params(csa_emulator *h, const csa_params *p, bool first)
{
    // On refresh (training) the previously packed buffers are simply re-filled in place: the
    // Uploader allocates on first use and the allocation order is deterministic.
    (void)first;
    const csa_config &c = h->dm.cfg;
    DevModel &d = h->dm;
    Uploader U{h};
    const int L = c.nlev, nin1 = c.nh1 + c.nh_mem;
    d.xmean_lev = U.up(p->xmean_lev, (size_t)L * c.nx);
    d.xdiv_lev = U.up(p->xdiv_lev, (size_t)L * c.nx);
    d.xmean_sca = U.up(p->xmean_sca, c.nx_sfc);
    d.xdiv_sca = U.up(p->xdiv_sca, c.nx_sfc);
    d.lbd_qc = U.up(p->lbd_qc, L);
    d.lbd_qi = U.up(p->lbd_qi, L);
    d.lbd_qn = c.v5_input ? U.up(p->lbd_qn, L) : nullptr;
    d.yscale_lev = U.up(p->yscale_lev, (size_t)L * c.ny);
    d.yscale_sca = U.up(p->yscale_sca, c.ny_sfc);
    d.hyam = U.up(p->hyam, L);
    d.hybm = U.up(p->hybm, L);
    d.init_wt = U.up(transposed(p->mlp_initial_w, c.nh1, c.nx + 1));
    d.init_b = U.up(p->mlp_initial_b, c.nh1);
    d.s1_wt = U.up(transposed(p->mlp_surface1_w, c.nh1, c.nx_sfc));
    d.s1_b = U.up(p->mlp_surface1_b, c.nh1);
    if (c.use_lstm) {
        d.s2_wt = U.up(transposed(p->mlp_surface2_w, c.nh1, c.nx_sfc));
        d.s2_b = U.up(p->mlp_surface2_b, c.nh1);
    }
    if (!c.legacy) {
        d.toa1_wt = U.up(transposed(p->mlp_toa1_w, c.nh2, 2));
        d.toa1_b = U.up(p->mlp_toa1_b, c.nh2);
        if (c.use_lstm) {
            d.toa2_wt = U.up(transposed(p->mlp_toa2_w, c.nh2, 2));
            d.toa2_b = U.up(p->mlp_toa2_b, c.nh2);
        }
    }
    // the two deterministic layers, in execution order: (rnn1, rnn2), or (rnn0, rnn1) for the stochastic variant
    const bool st = c.add_stochastic_layer != 0;
    const float *a_ih = st ? p->rnn0_w_ih : p->rnn1_w_ih, *a_hh = st ? p->rnn0_w_hh : p->rnn1_w_hh;
    const float *a_bi = st ? p->rnn0_b_ih : p->rnn1_b_ih, *a_bh = st ? p->rnn0_b_hh : p->rnn1_b_hh;
    const float *b_ih = st ? p->rnn1_w_ih : p->rnn2_w_ih, *b_hh = st ? p->rnn1_w_hh : p->rnn2_w_hh;
    const float *b_bi = st ? p->rnn1_b_ih : p->rnn2_b_ih, *b_bh = st ? p->rnn1_b_hh : p->rnn2_b_hh;
    std::vector<float> w, bias, bhn, packed;
    pack_ih(c.use_lstm, c.nh1, nin1, a_ih, a_bi, a_bh, w, bias, bhn);
    d.wih1 = U.up(w); d.bias1 = U.up(bias); d.bhn1 = U.up(bhn);
    pack_ih(c.use_lstm, c.nh2, c.nh1, b_ih, b_bi, b_bh, w, bias, bhn);
    d.wih2 = U.up(w); d.bias2 = U.up(bias); d.bhn2 = U.up(bhn);
    packed.resize(rec_packed_floats(c.use_lstm, c.nh1));
    rec_pack_weights(c.use_lstm, c.nh1, a_hh, packed.data());
    d.whh1p = U.up(packed);
    packed.resize(rec_packed_floats(c.use_lstm, c.nh2));
    rec_pack_weights(c.use_lstm, c.nh2, b_hh, packed.data());
    d.whh2p = U.up(packed);
    d.whh1q = d.whh2q = nullptr;
    if (c.use_lstm && c.nh1 <= 144 && c.nh2 <= 144) {     // one-column-per-workgroup latency kernel (small batches)
        packed.resize((size_t)4 * c.nh1 * c.nh1);
        rec1_pack_weights(c.nh1, a_hh, packed.data());
        d.whh1q = U.up(packed);
        packed.resize((size_t)4 * c.nh2 * c.nh2);
        rec1_pack_weights(c.nh2, b_hh, packed.data());
        d.whh2q = U.up(packed);
    }
    d.whh1m = d.whh2m = nullptr;
    if (c.use_lstm && c.nh1 <= 144 && c.nh2 <= 144 && c.nh1 % 16 == 0 && c.nh2 % 16 == 0) {     // matrix-pipe four-column kernel (large batches)
        packed.resize((size_t)4 * c.nh1 * c.nh1);
        rec4m_pack_weights(c.nh1, a_hh, packed.data());
        d.whh1m = U.up(packed);
        packed.resize((size_t)4 * c.nh2 * c.nh2);
        rec4m_pack_weights(c.nh2, b_hh, packed.data());
        d.whh2m = U.up(packed);
    }
    if (!c.use_lstm && c.nh1 <= 128 && c.nh2 <= 128 && c.nh1 % 16 == 0 && c.nh2 % 16 == 0) {     // GRU flavour of the matrix-pipe kernel
        packed.resize((size_t)4 * c.nh1 * c.nh1);
        gru4m_pack_weights(c.nh1, a_hh, packed.data());
        d.whh1m = U.up(packed);
        packed.resize((size_t)4 * c.nh2 * c.nh2);
        gru4m_pack_weights(c.nh2, b_hh, packed.data());
        d.whh2m = U.up(packed);
    }
    d.whh1g = d.whh2g = nullptr;
    if (!c.use_lstm && c.nh1 <= 144 && c.nh2 <= 144) {    // second-generation two-column GRU kernel
        packed.resize((size_t)3 * c.nh1 * c.nh1);
        gru2_pack_weights(c.nh1, a_hh, packed.data());
        d.whh1g = U.up(packed);
        packed.resize((size_t)3 * c.nh2 * c.nh2);
        gru2_pack_weights(c.nh2, b_hh, packed.data());
        d.whh2g = U.up(packed);
    }
    if (c.nh_mem > 0) {
        d.lat_wt = U.up(transposed(p->mlp_latent_w, c.nh_mem, c.nh2));
        d.lat_b = U.up(p->mlp_latent_b, c.nh_mem);
        d.out_w = U.up(p->mlp_output_w, (size_t)c.ny * c.nh_mem);
    } else {
        d.out_w = U.up(p->mlp_output_w, (size_t)c.ny * c.nh2);
        d.out_wt = U.up(transposed(p->mlp_output_w, c.ny, c.nh2));
    }
    d.out_b = U.up(p->mlp_output_b, c.ny);
    d.sfo_w = U.up(p->mlp_surface_output_w, (size_t)c.ny_sfc * c.nh2);
    d.sfo_b = U.up(p->mlp_surface_output_b, c.ny_sfc);
    return U.rc;
}

bool params_complete(const csa_config &c, const csa_params *p)
{
    bool ok = p->xmean_lev && p->xdiv_lev && p->xmean_sca && p->xdiv_sca && p->lbd_qc && p->lbd_qi &&
              p->yscale_lev && p->yscale_sca && p->hyam && p->hybm && p->mlp_initial_w && p->mlp_initial_b &&
              p->mlp_surface1_w && p->mlp_surface1_b && p->rnn1_w_ih && p->rnn1_w_hh && p->rnn1_b_ih &&
              p->rnn1_b_hh && p->mlp_output_w &&
              p->mlp_output_b && p->mlp_surface_output_w && p->mlp_surface_output_b;
    if (c.v5_input) ok = ok && p->lbd_qn;
    if (c.add_stochastic_layer) ok = ok && p->rnn0_w_ih && p->rnn0_w_hh && p->rnn0_b_ih && p->rnn0_b_hh && p->rnn2_weight_encoder;
    else ok = ok && p->rnn2_w_ih && p->rnn2_w_hh && p->rnn2_b_ih && p->rnn2_b_hh;
    if (c.use_lstm) ok = ok && p->mlp_surface2_w && p->mlp_surface2_b;
    if (!c.legacy) ok = ok && p->mlp_toa1_w && p->mlp_toa1_b && (!c.use_lstm || (p->mlp_toa2_w && p->mlp_toa2_b));
    if (c.nh_mem > 0) ok = ok && p->mlp_latent_w && p->mlp_latent_b;
    return ok;
}

void free_all(csa_emulator *h)
{
    for (void *p : h->owned) (void)hipFree(p);
    h->owned.clear();
}

}  // namespace

extern "C" int csa_create(const csa_config *cfg, const csa_params *hp, int max_batch, csa_emulator **out)
{
    if (!cfg || !hp || !out || max_batch <= 0) { csa_set_error_msg("csa_create: null argument"); return CSA_ERR_ARG; }
    int rc = check_cfg(*cfg);
    if (rc) return rc;
    if (!params_complete(*cfg, hp)) { csa_set_error_msg("csa_create: missing parameter array"); return CSA_ERR_ARG; }
    {
        // the register-stationary recurrent kernel is instantiated for these hidden sizes only
        const int ok1 = cfg->nh1 == 64 || cfg->nh1 == 96 || cfg->nh1 == 128 || cfg->nh1 == 144;
        const int ok2 = cfg->nh2 == 64 || cfg->nh2 == 96 || cfg->nh2 == 128 || cfg->nh2 == 144;
        if (!ok1 || !ok2) { csa_set_error_msg("csa_create: hidden size must be 64, 96, 128 or 144"); return CSA_ERR_UNSUPPORTED; }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        csa_set_error_msg("csa_create: no HIP device (the product path has no CPU fallback)");
        return CSA_ERR_HIP;
    }
    csa_emulator *h = new csa_emulator();
    memset(&h->dm, 0, sizeof(h->dm));
    h->dm.cfg = *cfg;
    h->max_batch = max_batch;
    rc = upload_params(h, hp, true);
    h->n_param_allocs = h->owned.size();     // everything allocated after this point is scratch (kept by csa_set_params)
    if (rc == CSA_OK) {
        Uploader U{h};
        const size_t L = cfg->nlev, Bm = max_batch;
        const size_t nhm = cfg->nh1 > cfg->nh2 ? cfg->nh1 : cfg->nh2;
        h->X1 = U.alloc(L * Bm * (cfg->nh1 + cfg->nh_mem));
        h->P = U.alloc(L * Bm * 4 * nhm);
        h->H1 = U.alloc(L * Bm * cfg->nh1);
        h->H2 = U.alloc(L * Bm * cfg->nh2);
        h->hc0 = U.alloc(4 * Bm * nhm);
        rc = U.rc;
        if (cfg->add_stochastic_layer) {
            h->ar_eps = U.alloc(L * Bm * cfg->nh2); h->ar_o6 = U.alloc(L * Bm * 6); h->ar_sfc = U.alloc(Bm * cfg->ny_sfc);
            h->ar_mem = U.alloc(L * Bm * cfg->nh_mem); h->ar_memT = U.alloc(L * Bm * cfg->nh_mem);
            rc = U.rc;
        }
        if (rc == CSA_OK && cfg->add_stochastic_layer)
            rc = csa_stoch_lstm4_create(cfg->nh1, cfg->nh2, hp->rnn2_weight_encoder, (int)(L * Bm), &h->stoch);
        if (rc == CSA_OK) {
            if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) rc = CSA_ERR_HIP;
            for (int i = 0; i < 2 && rc == CSA_OK; ++i)
                if (hipEventCreateWithFlags(&h->ov_ev[i], hipEventDisableTiming) != hipSuccess) rc = CSA_ERR_HIP;
        }
    }
    if (rc != CSA_OK) {
        free_all(h);
        if (h->stoch) (void)csa_stoch_destroy(h->stoch);
        delete h;
        if (g_err.empty()) csa_set_error_msg("csa_create: device allocation / upload failed");
        return rc;
    }
    *out = h;
    return CSA_OK;
}

extern "C" int csa_destroy(csa_emulator *h)
{
    if (!h) return CSA_ERR_ARG;
    free_all(h);
    if (h->stoch) (void)csa_stoch_destroy(h->stoch);
    for (int i = 0; i < 7; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    for (int i = 0; i < 2; ++i) if (h->ov_ev[i]) (void)hipEventDestroy(h->ov_ev[i]);
    if (h->side) (void)hipStreamDestroy(h->side);
    delete h;
    return CSA_OK;
}

extern "C" int csa_set_params(csa_emulator *h, const csa_params *hp)
{
    if (!h || !hp || !params_complete(h->dm.cfg, hp)) { csa_set_error_msg("csa_set_params: bad argument"); return CSA_ERR_ARG; }
    if (h->stoch) { csa_set_error_msg("csa_set_params: not available for the stochastic variant (re-create the handle)"); return CSA_ERR_UNSUPPORTED; }
    // Rebuild all parameter buffers; the scratch buffers (allocated after the parameters at creation) are kept.
    std::vector<void *> scratch(h->owned.begin() + h->n_param_allocs, h->owned.end());
    h->owned.resize(h->n_param_allocs);
    if (hipDeviceSynchronize() != hipSuccess) return CSA_ERR_HIP;
    free_all(h);
    int rc = upload_params(h, hp, false);
    h->n_param_allocs = h->owned.size();
    for (void *p : scratch) h->owned.push_back(p);
    return rc;
}

extern "C" int csa_packed_width(const csa_emulator *h)
{
    if (!h) return CSA_ERR_ARG;
    const csa_config &c = h->dm.cfg;
    return 6 * c.nlev + c.ny_sfc + c.nlev * c.nh_mem;
}
extern "C" int csa_max_batch(const csa_emulator *h) { return h ? h->max_batch : CSA_ERR_ARG; }
extern "C" const float *csa_tap_rnn1(const csa_emulator *h) { return h ? h->H1 : nullptr; }
extern "C" const float *csa_tap_rnn2(const csa_emulator *h) { return h ? h->H2 : nullptr; }

static const char *kStageNames[6] = {"prep", "proj_gemm_rnn1", "rec_rnn1", "proj_gemm_rnn2", "rec_rnn2", "head"};

static void prof_collect(csa_emulator *h)
{
    if (!h->pending) return;
    if (hipEventSynchronize(h->ev[6]) == hipSuccess) {
        for (int i = 0; i < 6; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]) == hipSuccess) h->acc_ms[i] += ms;
        }
        h->n_prof += 1;
    }
    h->pending = false;
}

extern "C" int csa_set_profiling(csa_emulator *h, int enable)
{
    if (!h) return CSA_ERR_ARG;
    if (enable && !h->ev[0])
        for (int i = 0; i < 7; ++i) CSA_HIP_CHECK(hipEventCreate(&h->ev[i]));
    if (!enable) prof_collect(h);
    h->profiling = enable != 0;
    return CSA_OK;
}

extern "C" int csa_reset_profile(csa_emulator *h)
{
    if (!h) return CSA_ERR_ARG;
    prof_collect(h);
    for (int i = 0; i < 6; ++i) h->acc_ms[i] = 0.0;
    h->n_prof = 0;
    return CSA_OK;
}

extern "C" int csa_get_profile(csa_emulator *h, double *avg_ms, int n, long *calls)
{
    if (!h || !avg_ms || n < 6) return CSA_ERR_ARG;
    prof_collect(h);
    for (int i = 0; i < 6; ++i) avg_ms[i] = h->n_prof ? h->acc_ms[i] / (double)h->n_prof : 0.0;
    if (calls) *calls = h->n_prof;
    return CSA_OK;
}

extern "C" const char *csa_stage_name(int i) { return (i >= 0 && i < 6) ? kStageNames[i] : ""; }

#define PROF_MARK(i)                                                 \
    do {                                                             \
        if (h->profiling) CSA_HIP_CHECK(hipEventRecord(h->ev[i], s)); \
    } while (0)

// Recurrent layer `layer` (1 or 2): up to 256 columns every column gets its own workgroup / CU (lstm_rec1_kernel, ~1.6x
// shorter steps); above that two columns share a workgroup (lstm_rec2_kernel), which is what saturates the chip.
static int launch_rec_auto(const csa_emulator *h, int layer, const float *P, const float *h0, const float *c0, float *Hout,
                           int B, int L, int reverse_out, hipStream_t s, int Bclass = 0)
{
    // Bclass: the batch the kernel class is chosen for (a column half inherits the class of the whole call, so that
    // the halves path stays bit-identical to the single-stream path)
    if (Bclass <= 0) Bclass = B;
    const DevModel &d = h->dm;
    const int nh = layer == 1 ? d.cfg.nh1 : d.cfg.nh2;
    const float *wq = layer == 1 ? d.whh1q : d.whh2q;
    if (d.cfg.use_lstm && wq && Bclass <= h->rec1_max_batch)
        return launch_rec1(nh, wq, P, h0, c0, Hout, B, L, reverse_out, s);
    const float *wm = layer == 1 ? d.whh1m : d.whh2m;
    if (wm && rec4m_selected(d.cfg.use_lstm, nh, Bclass))     // by the whole call's batch: column halves stay bit-identical
        return launch_rec4m(nh, wm, P, h0, c0, Hout, B, L, reverse_out, s);
    if (!d.cfg.use_lstm && Bclass <= h->rec1_max_batch)
        return launch_rec1_gru(nh, layer == 1 ? d.whh1p : d.whh2p, layer == 1 ? d.bhn1 : d.bhn2, P, h0, Hout, B, L, reverse_out, s);
    if (!d.cfg.use_lstm && wm && gru4m_selected(nh, Bclass))
        return launch_rec4m_gru(nh, wm, layer == 1 ? d.bhn1 : d.bhn2, P, h0, Hout, B, L, reverse_out, s);
    const float *wg = layer == 1 ? d.whh1g : d.whh2g;
    if (!d.cfg.use_lstm && wg)
        return launch_rec2_gru(nh, wg, layer == 1 ? d.bhn1 : d.bhn2, P, h0, Hout, B, L, reverse_out, s);
    if (!d.cfg.use_lstm) {     // (the first-generation kernel reads GRU projections padded to four per unit: training only)
        csa_set_error_msg("forward: no inference GRU kernel for this hidden size");
        return CSA_ERR_UNSUPPORTED;
    }
    return launch_rec(d.cfg.use_lstm, nh, layer == 1 ? d.whh1p : d.whh2p, layer == 1 ? d.bhn1 : d.bhn2, P, h0, c0, Hout, B, L,
                      reverse_out, s);
}

extern "C" int csa_set_rec1_max_batch(csa_emulator *h, int max_batch)
{
    if (!h || max_batch < 0) return CSA_ERR_ARG;
    h->rec1_max_batch = max_batch;
    return CSA_OK;
}

// Stochastic variant: LSTM down (noise init) -> LSTM up (surface init) -> stochastic LSTM down (TOA init).
// prep writes X1 in LEVEL order here (no flip), so the first recurrence runs downward and stores its hidden
// sequence flipped (= the sequence order of the upward pass), the second stores level order again.
static int run_forward_stoch(csa_emulator *h, int B, int normalised, int mode, const float *x_main, const float *x_sfc,
                             const float *mem_in, const float *hx0, const float *cx0, const float *eps,
                             float *y0, float *y1, float *y2, hipStream_t s)
{
    const csa_config &c = h->dm.cfg;
    if (B <= 0 || B > h->max_batch) { csa_set_error_msg("forward: B out of range (0 < B <= max_batch)"); return CSA_ERR_ARG; }
    if (!x_main || !x_sfc || !mem_in || !hx0 || !cx0 || !eps || !y0) { csa_set_error_msg("forward(noise): null tensor"); return CSA_ERR_ARG; }
    const int L = c.nlev, nh = c.nh1;
    int rc;
    if ((rc = launch_prep(h->dm, B, normalised, x_main, x_sfc, mem_in, nullptr, nullptr, h->X1, h->hc0, s))) return rc;
    if ((rc = launch_proj_gemm(h->X1, h->dm.wih1, h->dm.bias1, h->P, L * B, gate_stride(c.use_lstm) * nh, nh + c.nh_mem, s))) return rc;
    if ((rc = launch_rec_auto(h, 1, h->P, hx0, cx0, h->H1, B, L, /*reverse_out=*/1, s))) return rc;
    if ((rc = launch_proj_gemm(h->H1, h->dm.wih2, h->dm.bias2, h->P, L * B, gate_stride(c.use_lstm) * nh, nh, s))) return rc;
    if ((rc = launch_rec_auto(h, 2, h->P, h->hc0, h->hc0 + (size_t)B * nh, h->H2, B, L, /*reverse_out=*/1, s))) return rc;
    if ((rc = csa_stoch_lstm4_forward(h->stoch, L, B, h->H2, h->hc0 + (size_t)2 * B * nh, h->hc0 + (size_t)3 * B * nh, eps,
                                      h->H1, nullptr, nullptr, s))) return rc;
    return launch_head(h->dm, B, mode, h->H1, x_main, x_sfc, y0, y1, y2, s);
}

// Column-half concurrency: columns are independent, so the batch is cut in two
// and each half runs its own six launches on its own stream, with ONE fork and ONE join event.  While one half is in
// its recurrence (192 columns = 96 workgroups, VALU pipe, 2 of the 3 wave slots' registers) the other half's
// projection GEMM (MFMA pipe) can run on the same and on the idle CUs.  Scratch buffers are simply split in two.
static int run_chain(csa_emulator *h, int B, int normalised, int mode, const float *x_main, const float *x_sfc,
                     const float *mem_in, const float *hx2, const float *cx2, float *y0, float *y1, float *y2,
                     float *X1, float *P, float *H1, float *H2, float *hc0, int mem_B, int mem_off, hipStream_t s, int Bclass)
{
    const csa_config &c = h->dm.cfg;
    DevModel dm = h->dm;
    dm.mem_B = mem_B; dm.mem_off = mem_off;
    const int L = c.nlev;
    const size_t nhm = c.nh1 > c.nh2 ? c.nh1 : c.nh2;
    int rc;
    if ((rc = launch_prep(dm, B, normalised, x_main, x_sfc, mem_in, hx2, cx2, X1, hc0, s))) return rc;
    const float *h2 = c.legacy ? hx2 : hc0 + (size_t)2 * B * nhm;
    const float *c2 = c.legacy ? cx2 : hc0 + (size_t)3 * B * nhm;
    if ((rc = launch_proj_gemm(X1, h->dm.wih1, h->dm.bias1, P, L * B, gate_stride(c.use_lstm) * c.nh1, c.nh1 + c.nh_mem, s, L * Bclass))) return rc;
    if ((rc = launch_rec_auto(h, 1, P, hc0, hc0 + (size_t)B * nhm, H1, B, L, 1, s, Bclass))) return rc;
    if ((rc = launch_proj_gemm(H1, h->dm.wih2, h->dm.bias2, P, L * B, gate_stride(c.use_lstm) * c.nh2, c.nh1, s, L * Bclass))) return rc;
    if ((rc = launch_rec_auto(h, 2, P, h2, c2, H2, B, L, 0, s, Bclass))) return rc;
    return launch_head(dm, B, mode, H2, x_main, x_sfc, y0, y1, y2, s);
}

static int run_forward_halves(csa_emulator *h, int B, int normalised, int mode, const float *x_main, const float *x_sfc,
                              const float *mem_in, const float *hx2, const float *cx2, float *y0, float *y1, float *y2,
                              hipStream_t s)
{
    const csa_config &c = h->dm.cfg;
    const int L = c.nlev, B0 = ((B / 2) + 1) & ~1, B1 = B - B0;
    const size_t nhm = c.nh1 > c.nh2 ? c.nh1 : c.nh2, nin1 = c.nh1 + c.nh_mem;
    const int nxr = normalised ? c.nx : c.nx - (c.q_input_mode == 1);
    const int W = 6 * L + c.ny_sfc + L * c.nh_mem;
    hipStream_t T = h->side;
    int rc;
    CSA_HIP_CHECK(hipEventRecord(h->ov_ev[0], s));
    CSA_HIP_CHECK(hipStreamWaitEvent(T, h->ov_ev[0], 0));
    // level-major memory tensors (current generation) are addressed through (mem_B, mem_off); batch-first ones by pointer
    const bool lm = !c.legacy && c.nh_mem > 0;
    if ((rc = run_chain(h, B0, normalised, mode, x_main, x_sfc, mem_in, hx2, cx2, y0, y1, y2, h->X1, h->P, h->H1, h->H2, h->hc0,
                        lm ? B : 0, 0, s, B))) return rc;
    // second half: batch-first offsets of every caller tensor, the upper part of every scratch buffer
    const size_t o = B0;
    float *y0b = y0 + (mode == HEAD_PACKED ? o * W : o * L * (mode == HEAD_RAW || c.mp_mode == 0 ? c.ny : 6));
    rc = run_chain(h, B1, normalised, mode, x_main + o * L * nxr, x_sfc + o * c.nx_sfc,
                   mem_in ? (lm ? mem_in : mem_in + o * L * c.nh_mem) : nullptr,
                   hx2 ? hx2 + o * c.nh2 : nullptr, cx2 ? cx2 + o * c.nh2 : nullptr, y0b, y1 ? y1 + o * c.ny_sfc : nullptr,
                   y2 ? (lm ? y2 : y2 + o * L * c.nh_mem) : nullptr, h->X1 + (size_t)L * o * nin1, h->P + (size_t)L * o * 4 * nhm,
                   h->H1 + (size_t)L * o * c.nh1, h->H2 + (size_t)L * o * c.nh2, h->hc0 + 4 * o * nhm, lm ? B : 0, lm ? B0 : 0, T, B);
    if (rc) return rc;
    CSA_HIP_CHECK(hipEventRecord(h->ov_ev[1], T));
    CSA_HIP_CHECK(hipStreamWaitEvent(s, h->ov_ev[1], 0));
    return CSA_OK;
}

extern "C" int csa_set_halves(csa_emulator *h, int enable)
{
    if (!h) return CSA_ERR_ARG;
    if (enable < 0 || enable > 2) return CSA_ERR_ARG;
    h->halves = enable;
    return h->halves;
}

static int run_forward(csa_emulator *h, int B, int normalised, int mode,
                       const float *x_main, const float *x_sfc, const float *mem_in,
                       const float *hx2, const float *cx2,
                       float *y0, float *y1, float *y2, hipStream_t s)
{
    const csa_config &c = h->dm.cfg;
    if (c.add_stochastic_layer) { csa_set_error_msg("forward: the stochastic variant takes its noise explicitly (csa_*_noise entry points)"); return CSA_ERR_ARG; }
    if (B <= 0 || B > h->max_batch) { csa_set_error_msg("forward: B out of range (0 < B <= max_batch)"); return CSA_ERR_ARG; }
    if (!x_main || !x_sfc || !y0) { csa_set_error_msg("forward: null tensor"); return CSA_ERR_ARG; }
    if (c.nh_mem > 0 && !mem_in) { csa_set_error_msg("forward: rnn1_mem required (nh_mem > 0)"); return CSA_ERR_ARG; }
    if (c.legacy && (!hx2 || (c.use_lstm && !cx2))) { csa_set_error_msg("forward: legacy generation needs explicit hx2/cx2 noise"); return CSA_ERR_ARG; }
    const int L = c.nlev;
    const size_t nhm = c.nh1 > c.nh2 ? c.nh1 : c.nh2;
    int rc;
    // measured (tools/halves_sweep.py, memory wrapper): 0.90x at 384 columns (the co-running GEMM slows the latency-bound
    // recurrence of the other half), 1.04-1.09x from 768 columns up -> automatic from 640
    if ((h->halves == 1 || (h->halves == 2 && B >= 640)) && !h->profiling && B >= 64)
        return run_forward_halves(h, B, normalised, mode, x_main, x_sfc, mem_in, hx2, cx2, y0, y1, y2, s);
    if (h->profiling) prof_collect(h);   // previous profiled call (host sync: profiling mode only)
    PROF_MARK(0);
    if ((rc = launch_prep(h->dm, B, normalised, x_main, x_sfc, mem_in, hx2, cx2, h->X1, h->hc0, s))) return rc;
    const float *h2 = c.legacy ? hx2 : h->hc0 + (size_t)2 * B * nhm;
    const float *c2 = c.legacy ? cx2 : h->hc0 + (size_t)3 * B * nhm;
    PROF_MARK(1);
    // rnn1: upward over the flipped sequence; hidden sequence stored back in level order
    if ((rc = launch_proj_gemm(h->X1, h->dm.wih1, h->dm.bias1, h->P, L * B, gate_stride(c.use_lstm) * c.nh1, c.nh1 + c.nh_mem, s))) return rc;
    PROF_MARK(2);
    if ((rc = launch_rec_auto(h, 1, h->P, h->hc0, h->hc0 + (size_t)B * nhm, h->H1, B, L, /*reverse_out=*/1, s))) return rc;
    PROF_MARK(3);
    // rnn2: downward in level order
    if ((rc = launch_proj_gemm(h->H1, h->dm.wih2, h->dm.bias2, h->P, L * B, gate_stride(c.use_lstm) * c.nh2, c.nh1, s))) return rc;
    PROF_MARK(4);
    if ((rc = launch_rec_auto(h, 2, h->P, h2, c2, h->H2, B, L, /*reverse_out=*/0, s))) return rc;
    PROF_MARK(5);
    rc = launch_head(h->dm, B, mode, h->H2, x_main, x_sfc, y0, y1, y2, s);
    PROF_MARK(6);
    if (h->profiling) h->pending = true;
    return rc;
}

extern "C" int csa_forward_packed(csa_emulator *h, int B, const float *x_main, const float *x_sfc,
                                  const float *mem_in, const float *hx2, const float *cx2, float *yout, void *stream)
{
    if (!h) return CSA_ERR_ARG;
    if (h->dm.cfg.mp_mode != 1) { csa_set_error_msg("forward_packed: only the mp_mode 1 wrapper is packed (save_wrapper_mem.py:485-497)"); return CSA_ERR_UNSUPPORTED; }
    if (h->dm.cfg.nh_mem > 0 && !h->dm.cfg.legacy) { csa_set_error_msg("forward_packed: the current generation uses the tuple wrapper"); return CSA_ERR_UNSUPPORTED; }
    return run_forward(h, B, 0, HEAD_PACKED, x_main, x_sfc, mem_in, hx2, cx2, yout, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int csa_forward_tuple(csa_emulator *h, int B, const float *x_main, const float *x_sfc,
                                 const float *mem_in, float *out_lev, float *out_sfc, float *mem_out, void *stream)
{
    if (!h) return CSA_ERR_ARG;
    if (h->dm.cfg.legacy) { csa_set_error_msg("forward_tuple: current generation only"); return CSA_ERR_UNSUPPORTED; }
    if (!out_sfc || (h->dm.cfg.nh_mem > 0 && !mem_out)) { csa_set_error_msg("forward_tuple: null output"); return CSA_ERR_ARG; }
    return run_forward(h, B, 0, HEAD_TUPLE, x_main, x_sfc, mem_in, nullptr, nullptr, out_lev, out_sfc, mem_out, (hipStream_t)stream);
}

extern "C" int csa_forward_tuple_noise(csa_emulator *h, int B, const float *x_main, const float *x_sfc, const float *mem_in,
                                       const float *hx0, const float *cx0, const float *eps,
                                       float *out_lev, float *out_sfc, float *mem_out, void *stream)
{
    if (!h) return CSA_ERR_ARG;
    if (!h->stoch) { csa_set_error_msg("forward_tuple_noise: handle was not created with add_stochastic_layer"); return CSA_ERR_ARG; }
    if (!out_sfc || !mem_out) { csa_set_error_msg("forward_tuple_noise: null output"); return CSA_ERR_ARG; }
    return run_forward_stoch(h, B, 0, HEAD_TUPLE, x_main, x_sfc, mem_in, hx0, cx0, eps, out_lev, out_sfc, mem_out, (hipStream_t)stream);
}

extern "C" int csa_forward_packed_noise(csa_emulator *h, int B, const float *x_main, const float *x_sfc, const float *mem_in,
                                        const float *hx0, const float *cx0, const float *eps_prev, float *yout, void *stream)
{
    if (!h) return CSA_ERR_ARG;
    if (!h->stoch) { csa_set_error_msg("forward_packed_noise: handle was not created with add_stochastic_layer"); return CSA_ERR_ARG; }
    if (h->dm.cfg.mp_mode != 1) { csa_set_error_msg("forward_packed_noise: only the mp_mode 1 wrapper is packed"); return CSA_ERR_UNSUPPORTED; }
    if (!mem_in || !eps_prev || !yout || B <= 0 || B > h->max_batch) { csa_set_error_msg("forward_packed_noise: bad argument"); return CSA_ERR_ARG; }
    const csa_config &c = h->dm.cfg;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // the wrapper's two transposes (save_wrapper_mem.py:693 and the batch-first memory of the packed row)
    if ((rc = launch_to_level_major(B, c.nlev, c.nh2, eps_prev, h->ar_eps, s))) return rc;
    if ((rc = launch_to_level_major(B, c.nlev, c.nh_mem, mem_in, h->ar_memT, s))) return rc;
    if ((rc = run_forward_stoch(h, B, 0, HEAD_TUPLE, x_main, x_sfc, h->ar_memT, hx0, cx0, h->ar_eps, h->ar_o6, h->ar_sfc, h->ar_mem, s))) return rc;
    return launch_pack_ar(B, c.nlev, c.ny_sfc, c.nh_mem, c.nh2, h->ar_o6, h->ar_sfc, h->ar_mem, h->ar_eps, yout, s);
}

extern "C" int csa_model_forward_noise(csa_emulator *h, int B, const float *x_main_n, const float *x_sfc_n, const float *mem_in,
                                       const float *hx0, const float *cx0, const float *eps,
                                       float *out, float *out_sfc, float *mem_out, void *stream)
{
    if (!h) return CSA_ERR_ARG;
    if (!h->stoch) { csa_set_error_msg("model_forward_noise: handle was not created with add_stochastic_layer"); return CSA_ERR_ARG; }
    if (!out_sfc || !mem_out) { csa_set_error_msg("model_forward_noise: null output"); return CSA_ERR_ARG; }
    return run_forward_stoch(h, B, 1, HEAD_RAW, x_main_n, x_sfc_n, mem_in, hx0, cx0, eps, out, out_sfc, mem_out, (hipStream_t)stream);
}

extern "C" int csa_model_forward(csa_emulator *h, int B, const float *x_main_n, const float *x_sfc_n,
                                 const float *mem_in, const float *hx2, const float *cx2,
                                 float *out, float *out_sfc, float *mem_out, void *stream)
{
    if (!h) return CSA_ERR_ARG;
    if (!out_sfc || (h->dm.cfg.nh_mem > 0 && !mem_out)) { csa_set_error_msg("model_forward: null output"); return CSA_ERR_ARG; }
    return run_forward(h, B, 1, HEAD_RAW, x_main_n, x_sfc_n, mem_in, hx2, cx2, out, out_sfc, mem_out, (hipStream_t)stream);
}

extern "C" int csa_postprocess(csa_emulator *h, int B, const float *out, const float *out_sfc, const float *x_denorm, int nxd,
                               float *out_lev, float *out_sfc_denorm, void *stream)
{
    if (!h || B <= 0 || !out || !out_sfc || !x_denorm || !out_lev || !out_sfc_denorm) { csa_set_error_msg("csa_postprocess: bad argument"); return CSA_ERR_ARG; }
    if (h->dm.cfg.mp_mode == 0) { csa_set_error_msg("csa_postprocess: mp_mode 0 returns its inputs unchanged (models.py:278-279)"); return CSA_ERR_UNSUPPORTED; }
    return launch_postprocess(h->dm, B, out, out_sfc, x_denorm, nxd, out_lev, out_sfc_denorm, (hipStream_t)stream);
}

// ---- stage-wise entry point (parity evidence, tests/test_stagewise_parity.py) ------------------------------------------------
// Runs ONE stage of the six-launch path on caller-provided inputs in the library's internal layouts, so that every kernel can
// be fed the reference's exact input for that stage (teacher forcing) and compared with the reference's output of it:
//   0 prep      in0 x_main (B,L,nx) raw, in1 x_sfc (B,nx_sfc) raw, in2 mem_in (nullable), in3 hx2, in4 cx2 (legacy)
//               -> out0 X1 (L,B,nh1+nh_mem) SEQUENCE order (t = 0 is the surface), out1 hc0 (4,B,nhm)
//   1 / 3 proj  in0 X (L*B,K)            -> out0 P (L*B,4*nh), per unit [i,g~,f,o] (LSTM) / [r,z,n,0] (GRU), bias folded
//   2 / 4 rec   in0 P, in1 h0, in2 c0    -> out0 H (L,B,nh) LEVEL order (rnn1 runs upward: reverse_out = 1)
//   5 head      in0 H2 (L,B,nh2), in1 x_main raw, in2 x_sfc raw -> out0 packed (B,W)
//   6 / 7 step  in0 P (L*B,4*nh), in1 h_{t-1} (L*B,nh), in2 c_{t-1} -> out0 h_t (L*B,nh): ONE cell step per row (rnn1 / rnn2
//               weights), i.e. the recurrence kernel's arithmetic without the 60-step chain
// The kernel class (one- / two- / four-column recurrence, small / large GEMM) is the one a full call with this B would use.
extern "C" int csa_debug_stage(csa_emulator *h, int stage, int B, const float *in0, const float *in1, const float *in2,
                               const float *in3, const float *in4, float *out0, float *out1, void *stream)
{
    if (!h || B <= 0 || B > h->max_batch || !in0 || !out0) { csa_set_error_msg("csa_debug_stage: bad argument"); return CSA_ERR_ARG; }
    const csa_config &c = h->dm.cfg;
    if (c.add_stochastic_layer) { csa_set_error_msg("csa_debug_stage: not for the stochastic variant"); return CSA_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    const int L = c.nlev;
    switch (stage) {
    case 0:
        if (!in1 || !out1) { csa_set_error_msg("csa_debug_stage(prep): null tensor"); return CSA_ERR_ARG; }
        return launch_prep(h->dm, B, 0, in0, in1, in2, in3, in4, out0, out1, s);
    case 1: return launch_proj_gemm(in0, h->dm.wih1, h->dm.bias1, out0, L * B, gate_stride(c.use_lstm) * c.nh1, c.nh1 + c.nh_mem, s);
    case 3: return launch_proj_gemm(in0, h->dm.wih2, h->dm.bias2, out0, L * B, gate_stride(c.use_lstm) * c.nh2, c.nh1, s);
    case 2:
    case 4:
        if (!in1 || (c.use_lstm && !in2)) { csa_set_error_msg("csa_debug_stage(rec): null state"); return CSA_ERR_ARG; }
        return launch_rec_auto(h, stage == 2 ? 1 : 2, in0, in1, in2, out0, B, L, stage == 2 ? 1 : 0, s);
    case 6:
    case 7:     // ONE cell step for every (level, column) pair: nlev*B independent columns, sequence length 1
        if (!in1 || (c.use_lstm && !in2)) { csa_set_error_msg("csa_debug_stage(rec step): null state"); return CSA_ERR_ARG; }
        return launch_rec_auto(h, stage == 6 ? 1 : 2, in0, in1, in2, out0, L * B, 1, 0, s, B);
    case 5:
        if (!in1 || !in2) { csa_set_error_msg("csa_debug_stage(head): null tensor"); return CSA_ERR_ARG; }
        return launch_head(h->dm, B, HEAD_PACKED, in0, in1, in2, out0, nullptr, nullptr, s);
    default:
        csa_set_error_msg("csa_debug_stage: stage must be 0..7");
        return CSA_ERR_ARG;
    }
}
