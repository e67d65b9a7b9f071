/*
 * pxo_polya.c -- ORACLE (test infrastructure, never shipped): median
 * pre-filter, two-window t-statistic event detector, poly(A) interval search
 * and its retry / recalibration logic.
 *
 * Pinned against the real reference: tests/golden/events_*.npz come from the
 * reference's own C detector compiled as oracle/_ref/libscrappie_ref.so, and
 * tests/golden/polya_*.json from the imported poreplex/polya.py (py3.9).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "pxo.h"

/* scipy.signal.medfilt(x, k): zero padding at both ends, exact selection of
 * the middle order statistic (polya.py:62-63; also fast5_file.py:218). */
void pxo_medfilt(const float* x, int64_t n, int k, float* out)
{
    const int half = k / 2;
    float win[64];
    for (int64_t i = 0; i < n; i++) {
        for (int j = 0; j < k; j++) {
            const int64_t p = i - half + j;
            win[j] = (p >= 0 && p < n) ? x[p] : 0.0f;
        }
        /* insertion sort of a tiny window */
        for (int a = 1; a < k; a++) {
            float v = win[a];
            int b = a - 1;
            while (b >= 0 && win[b] > v) {
                win[b + 1] = win[b];
                b--;
            }
            win[b + 1] = v;
        }
        out[i] = win[half];
    }
}

/* ------------------------------------------------------------------------ *
 * a15  scrappie detect_events (event_detection.c:273-324) restated.
 *
 * prefix sums (:36-49): double accumulators; the square is a FLOAT product.
 * t-statistic (:62-117): trailing window in double, leading window in float,
 *   float means, variance clamped to FLT_MIN, double sqrt, float store.
 * peak FSM (:125-201): two detectors visited short-then-long per sample.
 * events (:217-271): boundaries at emitted peaks, float mean/stdv from the
 *   double prefix sums.
 * ------------------------------------------------------------------------ */
typedef struct {
    const float* t;      /* t-statistic trace             */
    float threshold;
    uint64_t window;
    uint64_t masked_to;
    int peak_pos;        /* -1 = none                     */
    float peak_val;
    int valid;
} detector_t;

static void tstat_trace(const double* cs, const double* cq, int64_t n, int64_t w,
                        float* t)
{
    for (int64_t i = 0; i < n; i++)
        t[i] = 0.0f;
    if (n < 2 * w || w < 2)
        return;
    const float wf = (float)w;
    for (int64_t i = w; i <= n - w; i++) {
        double s1 = cs[i], q1 = cq[i];
        if (i > w) {
            s1 -= cs[i - w];
            q1 -= cq[i - w];
        }
        const float s2 = (float)(cs[i + w] - cs[i]);
        const float q2 = (float)(cq[i + w] - cq[i]);
        const float m1 = (float)(s1 / wf);
        const float m2 = s2 / wf;
        const float m1sq = m1 * m1;
        const float m2sq = m2 * m2;
        const float q2w = q2 / wf;
        float var = (float)(((q1 / wf - (double)m1sq) + (double)q2w) - (double)m2sq);
        var = fmaxf(var, FLT_MIN);
        const float dm = m2 - m1;
        const float vw = var / wf;
        t[i] = (float)(fabs((double)dm) / sqrt((double)vw));
    }
}

int64_t pxo_detect_events(const float* sig, int64_t n, int64_t w1, int64_t w2,
                          float thr1, float thr2, float peak_height,
                          pxg_event* out, int64_t max_out)
{
    if (n <= 0)
        return 0;
    double* cs = (double*)malloc(sizeof(double) * (n + 1));
    double* cq = (double*)malloc(sizeof(double) * (n + 1));
    float* t1 = (float*)malloc(sizeof(float) * n);
    float* t2 = (float*)malloc(sizeof(float) * n);
    uint64_t* peaks = (uint64_t*)calloc(n, sizeof(uint64_t));
    cs[0] = 0.0;
    cq[0] = 0.0;
    for (int64_t i = 0; i < n; i++) {
        const float sq = sig[i] * sig[i];
        cs[i + 1] = cs[i] + (double)sig[i];
        cq[i + 1] = cq[i] + (double)sq;
    }
    tstat_trace(cs, cq, n, w1, t1);
    tstat_trace(cs, cq, n, w2, t2);

    detector_t det[2] = {
        { t1, thr1, (uint64_t)w1, 0, -1, FLT_MAX, 0 },
        { t2, thr2, (uint64_t)w2, 0, -1, FLT_MAX, 0 },
    };
    int64_t npk = 0;
    for (uint64_t i = 0; i < (uint64_t)n; i++) {
        for (int k = 0; k < 2; k++) {
            detector_t* d = &det[k];
            if (d->masked_to >= i)
                continue;
            const float cur = d->t[i];
            if (d->peak_pos == -1) {
                if (cur < d->peak_val) {
                    d->peak_val = cur;
                } else if (cur - d->peak_val > peak_height) {
                    d->peak_val = cur;
                    d->peak_pos = (int)i;
                }
            } else {
                if (cur > d->peak_val) {
                    d->peak_val = cur;
                    d->peak_pos = (int)i;
                }
                if (k == 0 && d->peak_val > d->threshold) {
                    det[1].masked_to = (uint64_t)d->peak_pos + d->window;
                    det[1].peak_pos = -1;
                    det[1].peak_val = FLT_MAX;
                    det[1].valid = 0;
                }
                if (d->peak_val - cur > peak_height && d->peak_val > d->threshold)
                    d->valid = 1;
                if (d->valid && (i - (uint64_t)d->peak_pos) > d->window / 2) {
                    peaks[npk++] = (uint64_t)d->peak_pos;
                    d->peak_pos = -1;
                    d->peak_val = cur;
                    d->valid = 0;
                }
            }
        }
    }
    /* event count = 1 + #{i: 0 < peaks[i] < n}  (:247-251) */
    int64_t ne = 1;
    for (int64_t i = 0; i < n; i++)
        if (peaks[i] > 0 && peaks[i] < (uint64_t)n)
            ne++;
    for (int64_t e = 0; e < ne && e < max_out; e++) {
        uint64_t b, en;
        if (e == 0) {
            b = 0;
            en = peaks[0];
        } else if (e < ne - 1) {
            b = peaks[e - 1];
            en = peaks[e];
        } else {
            b = peaks[ne - 2];
            en = (uint64_t)n;
        }
        pxg_event ev;
        ev.start = b;
        ev.length = (float)(en - b);
        ev.mean = (float)(cs[en] - cs[b]) / ev.length;
        const float dsq = (float)(cq[en] - cq[b]);
        const float msq = ev.mean * ev.mean;
        const float var = dsq / ev.length - msq;
        ev.stdv = sqrtf(fmaxf(var, 0.0f));
        ev.pos = -1;
        ev.state = -1;
        out[e] = ev;
    }
    free(cs); free(cq); free(t1); free(t2); free(peaks);
    return ne;
}

/* ------------------------------------------------------------------------ *
 * a16  polya.py:156-187  find_best_polya_interval: int64 DP, first maximum in
 *      row-major order of match*valid; <= 0 means "none".
 * ------------------------------------------------------------------------ */
int pxo_best_polya_interval(const pxg_config* cfg, const uint8_t* is_polya,
                            const float* length, int n, int* out_i, int* out_j)
{
    int64_t* w = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
    int64_t* sp = (int64_t*)malloc(sizeof(int64_t) * (n + 1));
    for (int k = 0; k < n; k++) {
        const double v = (is_polya[k] ? 1.0 : -1.0) * (double)length[k];
        w[k] = (int64_t)(v > 0 ? v : v * cfg->polya_spike_weight); /* trunc to 0 */
        sp[k] = is_polya[k] ? 1 : (int64_t)(-(double)length[k]);
    }
    /* np.argmax returns the FIRST maximum in row-major order; a maximum that
     * is <= 0 means "none", so only strictly positive cells can win and the
     * zero lower triangle never matters. */
    int64_t best = 0;
    int bi = 0, bj = 0;
    for (int i = 0; i < n; i++) {
        int64_t match = 0, spike = 0;
        for (int j = i; j < n; j++) {
            match += w[j];
            spike = spike < 0 ? -1
                  : (sp[j] > 0 ? (int64_t)cfg->polya_spike_tolerance : spike + sp[j]);
            const int64_t fin = spike > 0 ? match : 0;
            if (fin > best) {
                best = fin; bi = i; bj = j;
            }
        }
    }
    free(w); free(sp);
    if (best <= 0)
        return 0;
    *out_i = bi;
    *out_j = bj;
    return 1;
}

/* ---- a14/a17 driver ------------------------------------------------------ */
typedef struct {
    const pxg_config* cfg;
    const float* full;      /* scaled full-resolution signal */
    int64_t n_full;
    double sampling_rate;
    pxg_read_result* r;
    pxg_polya_spike* spikes;
    int spike_cap;                 /* rows the caller gave `spikes`; every spike is COUNTED */
} polya_ctx;

typedef struct {
    pxg_event* ev;
    int64_t* end;
    uint8_t* is_polya;
    int64_t n;
    float* sig;             /* filtered inspection window */
    int64_t n_sig;
} polya_win;

static void mark_polya(polya_win* W, double lo, double hi)
{
    /* Series(float32).between(lo, hi): bounds rounded to float32, inclusive */
    const float flo = (float)lo, fhi = (float)hi;
    for (int64_t k = 0; k < W->n; k++)
        W->is_polya[k] = (W->ev[k].mean >= flo) && (W->ev[k].mean <= fhi);
}

static void polya_entry(polya_ctx* C, int rough_begin, int rough_end, int has_end,
                        int has_range, double rlo, double rhi, int depth);

static void try_recalibrate(polya_ctx* C, polya_win* W, int64_t sig_begin,
                            int64_t sig_end, int base_b, int base_e,
                            int64_t adapter_end, int depth);

static float np_std_f32(const float* a, int64_t n)
{
    /* numpy _var/_std on float32: mean = fl(sum/n); sum of squared
     * deviations (pairwise float32); fl(ss/n); sqrtf */
    const float mean = pxo_np_sum_f32(a, n) / (float)n;
    float* d = (float*)malloc(sizeof(float) * n);
    for (int64_t i = 0; i < n; i++) {
        const float x = a[i] - mean;
        d[i] = x * x;
    }
    const float ss = pxo_np_sum_f32(d, n);
    free(d);
    return sqrtf(ss / (float)n);
}

static void call_polya(polya_ctx* C, polya_win* W, int64_t sig_begin, int64_t sig_end,
                       int base_b, int base_e, int64_t adapter_end, int has_range,
                       double rlo, double rhi, int depth)
{
    const pxg_config* cfg = C->cfg;
    const int stride = cfg->stride;
    float* len = (float*)malloc(sizeof(float) * W->n);
    for (int64_t k = 0; k < W->n; k++)
        len[k] = W->ev[k].length;
    int pi = 0, pj = -1;
    const int found = pxo_best_polya_interval(cfg, W->is_polya, len, (int)W->n, &pi, &pj);

    /* polya.py:81-85 open-ended: retry with a right-extended window */
    if (found && pj == W->n - 1 && sig_end < C->n_full &&
        depth < cfg->polya_maximum_openend_extension) {
        free(len);
        polya_entry(C, base_b, base_e + cfg->polya_openend_expansion / stride, 1,
                    has_range, rlo, rhi, depth + 1);
        return;
    }

    int shifted = 0;
    if (found && !has_range) {
        /* :88-93 length-weighted mean level, float32 reductions */
        const int m = pj - pi + 1;
        float* ml = (float*)malloc(sizeof(float) * m);
        for (int k = 0; k < m; k++)
            ml[k] = W->ev[pi + k].mean * len[pi + k];
        const float num = pxo_np_sum_f32(ml, m);
        const float den = pxo_np_sum_f32(len + pi, m);
        const float level = num / den;
        free(ml);
        shifted = fabs((double)level - cfg->polya_mean_dist[0]) >
                  cfg->polya_mean_trigger_recalibration * cfg->polya_mean_dist[1];
    }
    if (!found || shifted) {                                     /* :95-98 */
        free(len);
        try_recalibrate(C, W, sig_begin, sig_end, base_b, base_e, adapter_end, depth);
        return;
    }

    /* :101-102 longest event (first maximum) and its internal stdv */
    int lk = pi;
    for (int k = pi; k <= pj; k++)
        if (len[k] > len[lk])
            lk = k;
    const int64_t L = (int64_t)len[lk];
    const int64_t b = (int64_t)((double)W->ev[lk].start + (double)L * cfg->polya_stdv_range[0]);
    const int64_t e = (int64_t)((double)W->ev[lk].start + (double)L * cfg->polya_stdv_range[1]);
    int qc_ok = 0;
    if (e - b > 2) {
        /* signal[begin:end] clips to the window like a NumPy slice */
        int64_t bb = b < 0 ? 0 : b, ee = e > W->n_sig ? W->n_sig : e;
        if (ee > bb) {
            const float sd = np_std_f32(W->sig + bb, ee - bb);
            qc_ok = (double)sd < cfg->polya_stdv_max;
        }
    }
    if (qc_ok) {                                                  /* :104-121 */
        pxg_read_result* r = C->r;
        const int64_t pb = (int64_t)W->ev[pi].start;
        const int64_t pe = (int64_t)((double)W->ev[pj].start + (double)W->ev[pj].length);
        const int m = pj - pi + 1;
        float* pl = (float*)malloc(sizeof(float) * m);
        int np_ = 0, ns = 0;
        for (int k = pi; k <= pj; k++) {
            if (W->is_polya[k]) {
                pl[np_++] = len[k];
            } else {
                if (ns < C->spike_cap && C->spikes) {
                    C->spikes[ns].v[0] = len[k];
                    C->spikes[ns].v[1] = k - 1 >= pi ? W->ev[k - 1].mean : NAN;
                    C->spikes[ns].v[2] = W->ev[k].mean;
                    C->spikes[ns].v[3] = k + 1 <= pj ? W->ev[k + 1].mean : NAN;
                }
                ns++;
            }
        }
        const float dwell = pxo_np_sum_f32(pl, np_);
        free(pl);
        r->polya_called = 1;
        r->polya_begin = pb + sig_begin;
        r->polya_end = pe + sig_begin;
        r->polya_dwell_samples = (int32_t)dwell;
        r->polya_n_spikes = ns;            /* all of them (polya.py:109-115 lists every spike) */
        free(len);
        return;
    }
    free(len);
    if (!has_range)                                               /* :122-125 */
        try_recalibrate(C, W, sig_begin, sig_end, base_b, base_e, adapter_end, depth);
}

static void try_recalibrate(polya_ctx* C, polya_win* W, int64_t sig_begin,
                            int64_t sig_end, int base_b, int base_e,
                            int64_t adapter_end, int depth)
{
    const pxg_config* cfg = C->cfg;
    /* polya.py:131-136 anchor events near the adapter end */
    float* ml = (float*)malloc(sizeof(float) * (W->n > 0 ? W->n : 1));
    float* ll = (float*)malloc(sizeof(float) * (W->n > 0 ? W->n : 1));
    int m = 0;
    for (int64_t k = 0; k < W->n; k++) {
        if ((int64_t)W->ev[k].start <= adapter_end + cfg->recal_max_dist_from_adapter &&
            W->end[k] > adapter_end && (double)W->ev[k].stdv < cfg->recal_max_stdv) {
            ml[m] = W->ev[k].mean * W->ev[k].length;
            ll[m] = W->ev[k].length;
            m++;
        }
    }
    if (m == 0) {
        free(ml); free(ll);
        return;
    }
    const float pm = pxo_np_sum_f32(ml, m) / pxo_np_sum_f32(ll, m);  /* :138-139 */
    free(ml); free(ll);
    const double half = cfg->polya_mean_dist[1] * cfg->polya_mean_z_cutoff;
    const double lo = (double)pm - half, hi = (double)pm + half;       /* :140-142 */
    mark_polya(W, lo, hi);
    float* pl = (float*)malloc(sizeof(float) * (W->n > 0 ? W->n : 1));
    int np_ = 0;
    for (int64_t k = 0; k < W->n; k++)
        if (W->is_polya[k])
            pl[np_++] = W->ev[k].length;
    const float tot = pxo_np_sum_f32(pl, np_);
    free(pl);
    if ((double)tot >= (double)cfg->recal_min_length)                  /* :145 */
        call_polya(C, W, sig_begin, sig_end, base_b, base_e, adapter_end, 1, lo, hi, depth);
}

/* polya.py:50-73  __call__ */
static void polya_entry(polya_ctx* C, int rough_begin, int rough_end, int has_end,
                        int has_range, double rlo, double rhi, int depth)
{
    const pxg_config* cfg = C->cfg;
    const int stride = cfg->stride;
    const int min_unit = cfg->polya_openend_expansion / stride;
    const int rb = rough_begin;
    int re = rough_end;
    if (!has_end || re - rb < min_unit)
        re = rb + min_unit;
    int64_t ib = (int64_t)rb * stride - cfg->polya_refinement_expansion;
    if (ib < 0)
        ib = 0;
    int64_t ie = (int64_t)(re + 1) * stride + cfg->polya_refinement_expansion;
    if (ie > C->n_full)
        ie = C->n_full;
    const int64_t adapter_end = (int64_t)rb * stride - ib;
    const int64_t n = ie - ib;
    if (n <= 0)
        return;

    polya_win W;
    W.n_sig = n;
    W.sig = (float*)malloc(sizeof(float) * n);
    if (cfg->polya_median_pre_filter > 1)
        pxo_medfilt(C->full + ib, n, cfg->polya_median_pre_filter, W.sig);
    else
        memcpy(W.sig, C->full + ib, sizeof(float) * n);
    W.ev = (pxg_event*)malloc(sizeof(pxg_event) * (n + 1));
    W.n = pxo_detect_events(W.sig, n, cfg->ed_window_length1, cfg->ed_window_length2,
                            cfg->ed_threshold1, cfg->ed_threshold2, cfg->ed_peak_height,
                            W.ev, n + 1);
    W.end = (int64_t*)malloc(sizeof(int64_t) * (W.n > 0 ? W.n : 1));
    W.is_polya = (uint8_t*)malloc(W.n > 0 ? W.n : 1);
    for (int64_t k = 0; k < W.n; k++)   /* uint64 + float32 -> float64 -> int64 (:66) */
        W.end[k] = (int64_t)((double)W.ev[k].start + (double)W.ev[k].length);
    if (has_range) {
        mark_polya(&W, rlo, rhi);
    } else {
        const double half = cfg->polya_mean_dist[1] * cfg->polya_mean_z_cutoff;
        mark_polya(&W, cfg->polya_mean_dist[0] - half, cfg->polya_mean_dist[0] + half);
    }
    if (has_end)
        call_polya(C, &W, ib, ie, rb, re, adapter_end, has_range, rlo, rhi, depth);
    else
        try_recalibrate(C, &W, ib, ie, rb, re, adapter_end, depth);
    free(W.sig); free(W.ev); free(W.end); free(W.is_polya);
}

void pxo_polya(const pxg_config* cfg, const float* scaled_full, int64_t n_raw,
               int rough_begin, int rough_end_or_neg, double sampling_rate,
               pxg_read_result* r, pxg_polya_spike* spikes, int spike_cap)
{
    polya_ctx C = { cfg, scaled_full, n_raw, sampling_rate, r, spikes, spike_cap };
    r->polya_called = 0;
    r->polya_n_spikes = 0;
    r->polya_begin = r->polya_end = 0;
    r->polya_dwell_samples = 0;
    if (spikes) memset(spikes, 0, sizeof(pxg_polya_spike) * (size_t)(spike_cap > 0 ? spike_cap : 0));
    polya_entry(&C, rough_begin, rough_end_or_neg < 0 ? 0 : rough_end_or_neg,
                rough_end_or_neg >= 0, 0, 0.0, 0.0, 0);
}
