/*
 * climsim_oracle.c -- TEST INFRASTRUCTURE ONLY (see climsim_oracle.h).
 *
 * Plain-C, fp32, one column at a time.  Dot products accumulate in fp32 in
 * natural k order (k = 0..K-1), the same arithmetic class as the reference's
 * ATen/MKL fp32 path (whose internal summation order is unspecified).
 * Columns are independent, so the outer loop is an OpenMP parallel-for when
 * built with -fopenmp; results do not depend on the thread count.
 */
#include "climsim_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* y[o] = b[o] + sum_k Wt[k*O + o] * x[k]   (Wt = W transposed to (K,O)) */
static void matvec_t(const float *Wt, const float *b, const float *x, int K, int O, float *y)
{
    for (int o = 0; o < O; ++o) y[o] = b ? b[o] : 0.0f;
    for (int k = 0; k < K; ++k) {
        const float xv = x[k];
        const float *w = Wt + (size_t)k * O;
        for (int o = 0; o < O; ++o) y[o] += w[o] * xv;
    }
}

static float *transpose_new(const float *W, int O, int K)
{
    float *t = (float *)malloc(sizeof(float) * (size_t)O * K);
    if (!t) return NULL;
    for (int o = 0; o < O; ++o)
        for (int k = 0; k < K; ++k) t[(size_t)k * O + o] = W[(size_t)o * K + k];
    return t;
}

typedef struct {
    float *init_t, *s1_t, *s2_t, *toa1_t, *toa2_t;
    float *r1_ih_t, *r1_hh_t, *r2_ih_t, *r2_hh_t;
    float *lat_t, *outw_t, *sfo_t;
} tw_t;

static void tw_free(tw_t *t)
{
    free(t->init_t); free(t->s1_t); free(t->s2_t); free(t->toa1_t); free(t->toa2_t);
    free(t->r1_ih_t); free(t->r1_hh_t); free(t->r2_ih_t); free(t->r2_hh_t);
    free(t->lat_t); free(t->outw_t); free(t->sfo_t);
}

static int tw_build(const oracle_model *m, tw_t *t)
{
    const int G = m->use_lstm ? 4 : 3;
    const int nin1 = m->nh1 + m->nh_mem;
    memset(t, 0, sizeof(*t));
    t->init_t = transpose_new(m->mlp_initial_w, m->nh1, m->nx + 1);
    t->s1_t = transpose_new(m->mlp_surface1_w, m->nh1, m->nx_sfc);
    if (m->use_lstm) t->s2_t = transpose_new(m->mlp_surface2_w, m->nh1, m->nx_sfc);
    if (!m->legacy) {
        t->toa1_t = transpose_new(m->mlp_toa1_w, m->nh2, 2);
        if (m->use_lstm) t->toa2_t = transpose_new(m->mlp_toa2_w, m->nh2, 2);
    }
    t->r1_ih_t = transpose_new(m->rnn1_w_ih, G * m->nh1, nin1);
    t->r1_hh_t = transpose_new(m->rnn1_w_hh, G * m->nh1, m->nh1);
    t->r2_ih_t = transpose_new(m->rnn2_w_ih, G * m->nh2, m->nh1);
    t->r2_hh_t = transpose_new(m->rnn2_w_hh, G * m->nh2, m->nh2);
    if (m->nh_mem > 0) {
        t->lat_t = transpose_new(m->mlp_latent_w, m->nh_mem, m->nh2);
        t->outw_t = transpose_new(m->mlp_output_w, m->ny, m->nh_mem);
    } else {
        t->outw_t = transpose_new(m->mlp_output_w, m->ny, m->nh2);
    }
    t->sfo_t = transpose_new(m->mlp_surface_output_w, m->ny_sfc, m->nh2);
    return 0;
}

/* One LSTM/GRU step, PyTorch cell semantics.  gi = W_ih x + b_ih, gh = W_hh h + b_hh. */
static void cell_step(int use_lstm, int nh, const float *gi, const float *gh, float *h, float *c)
{
    if (use_lstm) {
        for (int j = 0; j < nh; ++j) {
            const float ig = sigmoidf_(gi[j] + gh[j]);
            const float fg = sigmoidf_(gi[nh + j] + gh[nh + j]);
            const float gg = tanhf(gi[2 * nh + j] + gh[2 * nh + j]);
            const float og = sigmoidf_(gi[3 * nh + j] + gh[3 * nh + j]);
            const float cn = fg * c[j] + ig * gg;
            c[j] = cn;
            h[j] = og * tanhf(cn);
        }
    } else {
        for (int j = 0; j < nh; ++j) {
            const float r = sigmoidf_(gi[j] + gh[j]);
            const float z = sigmoidf_(gi[nh + j] + gh[nh + j]);
            const float n = tanhf(gi[2 * nh + j] + r * gh[2 * nh + j]);
            h[j] = (1.0f - z) * n + z * h[j];
        }
    }
}

int oracle_model_forward(const oracle_model *m, int B,
                         const float *x_main_n, const float *x_sfc_n, const float *mem_in,
                         const float *hx2, const float *cx2,
                         float *out, float *out_sfc, float *mem_out,
                         float *rnn1out_dbg, float *rnn2out_dbg)
{
    const int L = m->nlev, nx = m->nx, nh1 = m->nh1, nh2 = m->nh2, nm = m->nh_mem;
    const int G = m->use_lstm ? 4 : 3;
    const int nin1 = nh1 + nm;
    if (m->legacy && (!hx2 || (m->use_lstm && !cx2))) return -2;
    if (nm > 0 && !mem_in) return -3;
    tw_t tw;
    tw_build(m, &tw);
    /* legacy artefacts hard-code the same numbers as xdiv_sca[0], xmean_sca[0] */
    const float sp_scale = m->xdiv_sca[0], sp_mean = m->xmean_sca[0];

#pragma omp parallel
    {
        float *x16 = (float *)malloc(sizeof(float) * (nx + 1));
        float *xin1 = (float *)malloc(sizeof(float) * (size_t)L * nin1); /* sequence order */
        float *r1 = (float *)malloc(sizeof(float) * (size_t)L * nh1);    /* level order   */
        float *r2 = (float *)malloc(sizeof(float) * (size_t)L * nh2);    /* level order   */
        float *gi = (float *)malloc(sizeof(float) * G * (nh1 > nh2 ? nh1 : nh2));
        float *gh = (float *)malloc(sizeof(float) * G * (nh1 > nh2 ? nh1 : nh2));
        float *h = (float *)malloc(sizeof(float) * (nh1 > nh2 ? nh1 : nh2));
        float *c = (float *)malloc(sizeof(float) * (nh1 > nh2 ? nh1 : nh2));
        float *b1 = (float *)malloc(sizeof(float) * G * nh1);
        float *lat = (float *)malloc(sizeof(float) * (nm > 0 ? nm : 1));
        float toa[2];
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            const float *xs = x_sfc_n + (size_t)b * m->nx_sfc;
            const float sp = xs[0] * sp_scale + sp_mean;
            /* rnn1 input in sequence order: t = 0 is the surface (level L-1) */
            for (int t = 0; t < L; ++t) {
                const int l = L - 1 - t;
                const float *xl = x_main_n + ((size_t)b * L + l) * nx;
                for (int k = 0; k < nx; ++k) x16[k] = xl[k];
                const float pres = m->hyam[l] * 100000.0f + sp * m->hybm[l];
                x16[nx] = sqrtf(pres) / 314.0f;
                float *xi = xin1 + (size_t)t * nin1;
                matvec_t(tw.init_t, m->mlp_initial_b, x16, nx + 1, nh1, xi);
                for (int j = 0; j < nh1; ++j) xi[j] = tanhf(xi[j]);
                if (nm > 0) {
                    /* legacy: memory indexed by sequence position (B,L,nm);
                     * current: level order, seq-first (L,B,nm), concatenated before the flip */
                    const float *mi = m->legacy ? mem_in + ((size_t)b * L + t) * nm
                                                : mem_in + ((size_t)l * B + b) * nm;
                    for (int j = 0; j < nm; ++j) xi[nh1 + j] = mi[j];
                }
            }
            /* initial state from the surface MLPs */
            matvec_t(tw.s1_t, m->mlp_surface1_b, xs, m->nx_sfc, nh1, h);
            for (int j = 0; j < nh1; ++j) h[j] = tanhf(h[j]);
            if (m->use_lstm) {
                matvec_t(tw.s2_t, m->mlp_surface2_b, xs, m->nx_sfc, nh1, c);
                if (m->legacy) for (int j = 0; j < nh1; ++j) c[j] = tanhf(c[j]);
            }
            /* rnn1 upward */
            for (int t = 0; t < L; ++t) {
                matvec_t(tw.r1_ih_t, m->rnn1_b_ih, xin1 + (size_t)t * nin1, nin1, G * nh1, gi);
                matvec_t(tw.r1_hh_t, m->rnn1_b_hh, h, nh1, G * nh1, gh);
                cell_step(m->use_lstm, nh1, gi, gh, h, c);
                memcpy(r1 + (size_t)(L - 1 - t) * nh1, h, sizeof(float) * nh1);
            }
            /* rnn2 initial state */
            if (m->legacy) {
                memcpy(h, hx2 + (size_t)b * nh2, sizeof(float) * nh2);
                if (m->use_lstm) memcpy(c, cx2 + (size_t)b * nh2, sizeof(float) * nh2);
            } else {
                toa[0] = xs[1]; toa[1] = xs[6];
                matvec_t(tw.toa1_t, m->mlp_toa1_b, toa, 2, nh2, h);
                if (m->use_lstm) matvec_t(tw.toa2_t, m->mlp_toa2_b, toa, 2, nh2, c);
            }
            /* rnn2 downward (level order) */
            for (int l = 0; l < L; ++l) {
                matvec_t(tw.r2_ih_t, m->rnn2_b_ih, r1 + (size_t)l * nh1, nh1, G * nh2, gi);
                matvec_t(tw.r2_hh_t, m->rnn2_b_hh, h, nh2, G * nh2, gh);
                cell_step(m->use_lstm, nh2, gi, gh, h, c);
                memcpy(r2 + (size_t)l * nh2, h, sizeof(float) * nh2);
            }
            /* heads */
            for (int l = 0; l < L; ++l) {
                float *o = out + ((size_t)b * L + l) * m->ny;
                if (nm > 0) {
                    matvec_t(tw.lat_t, m->mlp_latent_b, r2 + (size_t)l * nh2, nh2, nm, lat);
                    float *mo = m->legacy ? mem_out + ((size_t)b * L + (L - 1 - l)) * nm
                                          : mem_out + ((size_t)l * B + b) * nm;
                    memcpy(mo, lat, sizeof(float) * nm);
                    matvec_t(tw.outw_t, m->mlp_output_b, lat, nm, m->ny, o);
                } else {
                    matvec_t(tw.outw_t, m->mlp_output_b, r2 + (size_t)l * nh2, nh2, m->ny, o);
                }
                if (m->output_prune && l < 12)
                    for (int v = 1; v < m->ny; ++v) o[v] = 0.0f;
            }
            matvec_t(tw.sfo_t, m->mlp_surface_output_b, h, nh2, m->ny_sfc, out_sfc + (size_t)b * m->ny_sfc);
            if (rnn1out_dbg) memcpy(rnn1out_dbg + (size_t)b * L * nh1, r1, sizeof(float) * L * nh1);
            if (rnn2out_dbg) memcpy(rnn2out_dbg + (size_t)b * L * nh2, r2, sizeof(float) * L * nh2);
        }
        free(x16); free(xin1); free(r1); free(r2); free(gi); free(gh); free(h); free(c); free(b1); free(lat);
    }
    tw_free(&tw);
    return 0;
}

/* rnn/utils.py:134-180: saturation vapour pressure polynomials and RH -> specific humidity */
static float polyval9(const float *a, float x)
{
    float o = 0.0f;
    for (int i = 0; i < 9; ++i) o = o * x + a[i];
    return o;
}
static float rh_to_q(float rh, float T, float p)
{
    static const float a_liq[9] = {-0.976195544e-15f, -0.952447341e-13f, 0.640689451e-10f, 0.206739458e-7f,
                                   0.302950461e-5f, 0.264847430e-3f, 0.142986287e-1f, 0.443987641f, 6.11239921f};
    static const float a_ice[9] = {0.252751365e-14f, 0.146898966e-11f, 0.385852041e-9f, 0.602588177e-7f,
                                   0.615021634e-5f, 0.420895665e-3f, 0.188439774e-1f, 0.503160820f, 6.11147274f};
    const float T0 = 273.16f;
    float xl = T - T0; if (xl < -80.0f) xl = -80.0f;
    const float eliq = 100.0f * polyval9(a_liq, xl);
    float eice;
    if (T > 273.15f) eice = eliq;
    else if (T > 185.0f) eice = 100.0f * polyval9(a_ice, T - T0);
    else {
        float tmp = T - T0; if (tmp < -100.0f) tmp = -100.0f;
        eice = 100.0f * (0.00763685f + tmp * (0.000151069f + tmp * 7.48215e-07f));
    }
    float omega = (T - 253.16f) / 20.0f;
    omega = omega < 0.0f ? 0.0f : (omega > 1.0f ? 1.0f : omega);
    const float esat = omega * eliq + (1.0f - omega) * eice;
    return rh * ((287.0f * esat) / (461.0f * p));
}

int oracle_preprocess(const oracle_model *m, int B, const float *x_main, const float *x_sfc,
                      float *x_main_n, float *x_sfc_n)
{
    const int L = m->nlev, nx = m->nx;
    const int nxr = nx - (m->q_input_mode == 1);      /* columns of the raw input */
    for (int b = 0; b < B; ++b) {
        for (int l = 0; l < L; ++l) {
            const float *xi = x_main + ((size_t)b * L + l) * nxr;
            float *xo = x_main_n + ((size_t)b * L + l) * nx;
            for (int v = 0; v < nx; ++v) {
                float x = v < nxr ? xi[v] : 0.0f;
                if ((m->q_input_mode == 1 && v == nxr) || (m->q_input_mode == 2 && v == 1)) {
                    const float pres = m->hyam[l] * 100000.0f + x_sfc[(size_t)b * m->nx_sfc] * m->hybm[l];
                    x = rh_to_q(xi[1], xi[0], pres);
                }
                if (m->v5_input) {          /* rnn/utils.py:186-198 */
                    if (v == 2) {
                        x = xi[2] + xi[3];
                        if (m->qinput_prune && l < 15) x = 0.0f;
                        x = 1.0f - expf(-x * m->lbd_qn[l]);
                    }
                    if (v == 3) {           /* temperature_scaling, models.py:260-266 */
                        x = (xi[0] - 253.16f) * 0.05f;
                        x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
                    }
                } else {
                    if (v == 2) x = 1.0f - expf(-x * m->lbd_qc[l]);
                    if (v == 3) x = 1.0f - expf(-x * m->lbd_qi[l]);
                }
                x = (x - m->xmean_lev[l * nx + v]) / m->xdiv_lev[l * nx + v];
                if (!m->v5_input && m->qinput_prune && v == 2 && l < 15) x = 0.0f;
                if (m->rh_prune && v == 1) {
                    /* torch.clamp propagates NaN */
                    if (!isnan(x)) x = x < 0.0f ? 0.0f : (x > 1.2f ? 1.2f : x);
                }
                if (isnan(x)) x = 0.0f;
                if (m->scrub_inf && isinf(x)) x = 0.0f;
                xo[v] = x;
            }
        }
        for (int v = 0; v < m->nx_sfc; ++v) {
            float x = x_sfc[(size_t)b * m->nx_sfc + v];
            if (m->snowhice_fix && x >= 1e10f) x = -1.0f;
            x_sfc_n[(size_t)b * m->nx_sfc + v] = (x - m->xmean_sca[v]) / m->xdiv_sca[v];
        }
    }
    return 0;
}

/* microphysics partition shared by both wrappers (save_wrapper_mem.py:470-483, models.py:305-329) */
static inline void mp_partition(float T_old, float qliq_old, float qice_old, float dT, float dqn,
                                float *dqliq, float *dqice)
{
    const float T_new = T_old + dT * 1200.0f;
    float lf = (T_new - 253.16f) * 0.05f;
    /* F.hardtanh(x,0,1) == clamp; NaN propagates */
    if (!isnan(lf)) lf = lf < 0.0f ? 0.0f : (lf > 1.0f ? 1.0f : lf);
    const float qn_old = qliq_old + qice_old;
    const float qn_new = qn_old + dqn * 1200.0f;
    const float qliq_new = lf * qn_new;
    const float qice_new = (1.0f - lf) * qn_new;
    *dqliq = (qliq_new - qliq_old) * 0.0008333333333333334f;
    *dqice = (qice_new - qice_old) * 0.0008333333333333334f;
}

int oracle_wrapper_forward(const oracle_model *m, int B,
                           const float *x_main, const float *x_sfc, const float *mem_in,
                           const float *hx2, const float *cx2, float *yout)
{
    const int L = m->nlev, nx = m->nx, ny = m->ny, nm = m->nh_mem;
    const int nout = 6 * L + m->ny_sfc + L * nm;
    if (m->mp_mode != 1 || ny != 5) return -4;
    float *xn = (float *)malloc(sizeof(float) * (size_t)B * L * nx);
    float *xsn = (float *)malloc(sizeof(float) * (size_t)B * m->nx_sfc);
    float *out = (float *)malloc(sizeof(float) * (size_t)B * L * ny);
    float *osfc = (float *)malloc(sizeof(float) * (size_t)B * m->ny_sfc);
    float *memo = nm > 0 ? (float *)malloc(sizeof(float) * (size_t)B * L * nm) : NULL;
    oracle_preprocess(m, B, x_main, x_sfc, xn, xsn);
    int rc = oracle_model_forward(m, B, xn, xsn, mem_in, hx2, cx2, out, osfc, memo, NULL, NULL);
    if (rc == 0) {
        for (int b = 0; b < B; ++b) {
            float *y = yout + (size_t)b * nout;
            for (int l = 0; l < L; ++l) {
                const float *o = out + ((size_t)b * L + l) * ny;
                const float *ys = m->yscale_lev + (size_t)l * ny;
                const float *xr = x_main + ((size_t)b * L + l) * nx;
                const float dT = o[0] / ys[0], dqv = o[1] / ys[1], dqn = o[2] / ys[2];
                const float du = o[3] / ys[3], dv = o[4] / ys[4];
                float dql, dqi;
                mp_partition(xr[0], xr[2], xr[3], dT, dqn, &dql, &dqi);
                y[0 * L + l] = dT;  y[1 * L + l] = dqv;
                y[2 * L + l] = dql; y[3 * L + l] = dqi;
                y[4 * L + l] = du;  y[5 * L + l] = dv;
            }
            for (int v = 0; v < m->ny_sfc; ++v)
                y[6 * L + v] = osfc[(size_t)b * m->ny_sfc + v] / m->yscale_sca[v];
            if (nm > 0) {
                /* legacy (B,L,nm) block copied as-is; current generation's packed wrapper
                 * (save_wrapper_mem.py:497) reshapes whatever layout the model returned */
                const float *src = memo + (size_t)b * L * nm;
                if (m->legacy) memcpy(y + 6 * L + m->ny_sfc, src, sizeof(float) * L * nm);
                else return -5;
            }
            if (m->scrub_out_nan)
                for (int i = 0; i < nout; ++i) if (isnan(y[i])) y[i] = 0.0f;
        }
    }
    free(xn); free(xsn); free(out); free(osfc); free(memo);
    return rc;
}

int oracle_wrapper_forward_tuple(const oracle_model *m, int B,
                                 const float *x_main, const float *x_sfc, const float *mem_in,
                                 float *out_lev, float *out_sfc, float *mem_out)
{
    const int L = m->nlev, nx = m->nx, ny = m->ny;
    if (m->legacy) return -6;
    const int nxr = nx - (m->q_input_mode == 1);
    float *xn = (float *)malloc(sizeof(float) * (size_t)B * L * nx);
    float *xsn = (float *)malloc(sizeof(float) * (size_t)B * m->nx_sfc);
    float *out = (float *)malloc(sizeof(float) * (size_t)B * L * ny);
    float *osfc = (float *)malloc(sizeof(float) * (size_t)B * m->ny_sfc);
    oracle_preprocess(m, B, x_main, x_sfc, xn, xsn);
    int rc = oracle_model_forward(m, B, xn, xsn, mem_in, NULL, NULL, out, osfc, mem_out, NULL, NULL);
    if (rc == 0) {
        if (m->mp_mode == 0) {
            /* models.py:278-279: mp_mode 0 returns the UN-denormalised outputs */
            memcpy(out_lev, out, sizeof(float) * (size_t)B * L * ny);
            memcpy(out_sfc, osfc, sizeof(float) * (size_t)B * m->ny_sfc);
        } else if (m->mp_mode == 1) {
            for (int b = 0; b < B; ++b) {
                for (int l = 0; l < L; ++l) {
                    const float *o = out + ((size_t)b * L + l) * ny;
                    const float *ys = m->yscale_lev + (size_t)l * ny;
                    const float *xr = x_main + ((size_t)b * L + l) * nxr;
                    float *y = out_lev + ((size_t)b * L + l) * 6;
                    const float dT = o[0] / ys[0], dqn = o[2] / ys[2];
                    float dql, dqi;
                    mp_partition(xr[0], xr[2], xr[3], dT, dqn, &dql, &dqi);
                    y[0] = dT; y[1] = o[1] / ys[1]; y[2] = dql; y[3] = dqi;
                    y[4] = o[3] / ys[3]; y[5] = o[4] / ys[4];
                    /* rnn/utils.py:290: NaN -> 0 on out_lev only */
                    for (int v = 0; v < 6; ++v) if (isnan(y[v])) y[v] = 0.0f;
                }
                for (int v = 0; v < m->ny_sfc; ++v)
                    out_sfc[(size_t)b * m->ny_sfc + v] = osfc[(size_t)b * m->ny_sfc + v] / m->yscale_sca[v];
            }
        } else {
            rc = -4;
        }
        if (m->mp_mode == 0)
            for (size_t i = 0; i < (size_t)B * L * ny; ++i) if (isnan(out_lev[i])) out_lev[i] = 0.0f;
    }
    free(xn); free(xsn); free(out); free(osfc);
    return rc;
}
