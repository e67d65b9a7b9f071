/*
 * pxo_unsplit.c -- ORACLE (test infrastructure, never shipped): the Guppy
 * event table (a18) and the window scan of the pseudo-fusion filter (a19).
 * Pinned by the real fast5_file.py / signal_analyzer.py glue (py3.9) in
 * tests/golden/chimera.*; the Viterbi inside the scan is the restated
 * pomegranate semantics (parity unpinned, see pxo.h).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "pxo.h"

int pxo_guppy_event_means(const int16_t* raw, int64_t n_raw, const pxg_calib* cal,
                          int64_t first, int64_t n_events, int stride, float scale,
                          float shift, float* mean, float* scaled)
{
    return pxo_guppy_event_table(raw, n_raw, cal, first, n_events, stride, scale, shift, mean, NULL, scaled);
}

/* + the per-event standard deviation of convert_events_guppy (fast5_file.py:227,
 * sigbyevents.std(axis=1) on float32: NumPy's _var -- the row mean, fl(x - mean), fl(d * d),
 * the same pairwise row sum, / n, sqrt, every step rounded to float32) */
int pxo_guppy_event_table(const int16_t* raw, int64_t n_raw, const pxg_calib* cal,
                          int64_t first, int64_t n_events, int stride, float scale,
                          float shift, float* mean, float* stdv_or_null, float* scaled)
{
    /* fast5_file.py:212-218: rawdata = get_raw_data(first, last) (end clamped
     * to the dataset), medfilt(5) with zero padding */
    const int64_t last = first + (int64_t)stride * n_events;
    int64_t end = last > n_raw ? n_raw : last;
    int64_t len = end - first;
    if (len < 0) len = 0;
    float* pa = (float*)malloc(sizeof(float) * (len > 0 ? len : 1));
    float* filt = (float*)malloc(sizeof(float) * (len > 0 ? len : 1));
    if (len > 0) {
        pxo_raw_to_pa(raw + first, len, cal, pa);
        pxo_medfilt(pa, len, 5, filt);
    }
    /* :219-223 NaN padding to a multiple of the stride, then the count check */
    int64_t padded = len;
    if (len % stride > 0)
        padded = len + (stride - len % stride);
    if (padded / stride != n_events) {
        free(pa); free(filt);
        return -1;
    }
    for (int64_t k = 0; k < n_events; k++) {
        float blk[128];
        int has_nan = 0;
        for (int j = 0; j < stride; j++) {
            const int64_t q = k * stride + j;
            if (q < len) blk[j] = filt[q];
            else { blk[j] = NAN; has_nan = 1; }
        }
        (void)has_nan;
        float s = 0.0f + pxo_np_sum_f32(blk, stride);     /* :226 mean(axis=1) */
        const float m = s / (float)stride;
        mean[k] = m;
        const float y = scale * m;                         /* signal_analyzer.py:318 */
        scaled[k] = y + shift;
        if (stdv_or_null) {
            float sq[128];
            for (int j = 0; j < stride; j++) {
                const float d = blk[j] - m;
                sq[j] = d * d;
            }
            const float v = (0.0f + pxo_np_sum_f32(sq, stride)) / (float)stride;
            stdv_or_null[k] = sqrtf(v);
        }
    }
    free(pa); free(filt);
    return 0;
}

int pxo_unsplit_scan(const pxg_config* cfg, const float* scaled_mean, int64_t n_events,
                     int64_t first, int stride, int64_t payload_start, double rate,
                     int64_t* intervals, int cap)
{
    const pxg_hmm* hmm = &cfg->unsplit_model;
    const int A = hmm->adapter_state, LL = hmm->leader_low_state, LH = hmm->leader_high_state;
    /* signal_analyzer.py:374-383: int(config[name] * sampling_rate) */
    const int64_t window_size = (int64_t)(cfg->unsplit_window_size * rate);
    const int64_t window_step = (int64_t)(cfg->unsplit_window_step * rate);
    const int64_t strict_duration = (int64_t)(cfg->unsplit_strict_duration * rate);
    const int64_t cut_total[2] = { (int64_t)(cfg->unsplit_loosen_full_length * rate),
                                   (int64_t)(cfg->unsplit_strict_full_length * rate) };
    const int64_t cut_adapter[2] = { (int64_t)(cfg->unsplit_loosen_dna_length * rate),
                                     (int64_t)(cfg->unsplit_strict_dna_length * rate) };
    if (n_events <= 0 || window_step <= 0)
        return 0;
    /* event k: start = first + stride*k; end = next start, last event: start+1
     * (signal_analyzer.py:321-324) */
    const int64_t last_end = first + (int64_t)stride * (n_events - 1) + 1;
    int count = 0;
    int32_t* path = (int32_t*)malloc(sizeof(int32_t) * (n_events > 0 ? n_events : 1));
    for (int64_t left = payload_start; left < last_end; left += window_step) {   /* :385 */
        /* events with left <= start <= left + window_size (inclusive both ends) */
        int64_t k0 = left - first <= 0 ? 0 : (left - first + stride - 1) / stride;
        int64_t k1 = left + window_size - first < 0 ? -1 : (left + window_size - first) / stride;
        if (k1 > n_events - 1)
            k1 = n_events - 1;
        if (k1 < k0)
            break;                                                                /* :387-388 */
        const int T = (int)(k1 - k0 + 1);
        pxo_viterbi(hmm, scaled_mean + k0, T, path);
        int leader_start = -1;
        int t = 0;
        while (t < T) {                                                           /* :393-418 */
            int e = t;
            while (e + 1 < T && path[e + 1] == path[t])
                e++;
            const int st = path[t];
            if (st != A && st != LH && st != LL) {
                leader_start = -1;
            } else {
                if (leader_start < 0)
                    leader_start = t;
                if (st == A) {
                    const int64_t ev_last = k0 + e, ev_lead = k0 + leader_start, ev_first = k0 + t;
                    const int64_t adapter_end = (ev_last == n_events - 1)
                        ? first + stride * ev_last + 1 : first + stride * (ev_last + 1);
                    const int64_t leader_in_read = first + stride * ev_lead;
                    const int64_t total_duration = adapter_end - leader_in_read;
                    const int64_t adapter_duration = adapter_end - (first + stride * ev_first);
                    const int strict = (leader_in_read - payload_start) <= strict_duration;
                    if (total_duration >= cut_total[strict] && adapter_duration >= cut_adapter[strict]) {
                        if (count < cap) {
                            intervals[2 * count] = leader_in_read;
                            intervals[2 * count + 1] = 1 + adapter_end;
                        }
                        count++;
                    }
                    leader_start = -1;
                }
            }
            t = e + 1;
        }
    }
    free(path);
    return count;
}

/* a19 over a table that brings its own events (albacore's 14-column Events, fast5_file.py:178-179):
 * starts[k] = the table's `start' column (ascending), an event ends where the next starts, the last one a
 * sample after its start (signal_analyzer.py:321-324); the window of `left' holds the events with
 * left <= start <= left + window_size (:384-386: Series.between, both ends inclusive); everything else as
 * pxo_unsplit_scan.  scaled_mean = fl(fl(scale * mean) + shift) (:318) is the caller's. */
int pxo_unsplit_scan_events(const pxg_config* cfg, const float* scaled_mean, const int64_t* starts,
                            int64_t n_events, int64_t payload_start, double rate, int64_t* intervals, int cap)
{
    const pxg_hmm* hmm = &cfg->unsplit_model;
    const int A = hmm->adapter_state, LL = hmm->leader_low_state, LH = hmm->leader_high_state;
    const int64_t window_size = (int64_t)(cfg->unsplit_window_size * rate);
    const int64_t window_step = (int64_t)(cfg->unsplit_window_step * rate);
    const int64_t strict_duration = (int64_t)(cfg->unsplit_strict_duration * rate);
    const int64_t cut_total[2] = { (int64_t)(cfg->unsplit_loosen_full_length * rate),
                                   (int64_t)(cfg->unsplit_strict_full_length * rate) };
    const int64_t cut_adapter[2] = { (int64_t)(cfg->unsplit_loosen_dna_length * rate),
                                     (int64_t)(cfg->unsplit_strict_dna_length * rate) };
    if (n_events <= 0 || window_step <= 0)
        return 0;
    const int64_t last_end = starts[n_events - 1] + 1;
    int count = 0;
    int32_t* path = (int32_t*)malloc(sizeof(int32_t) * n_events);
    for (int64_t left = payload_start; left < last_end; left += window_step) {
        int64_t k0 = 0, k1 = n_events - 1;
        while (k0 < n_events && starts[k0] < left) k0++;
        while (k1 >= 0 && starts[k1] > left + window_size) k1--;
        if (k1 < k0)
            break;
        const int T = (int)(k1 - k0 + 1);
        pxo_viterbi(hmm, scaled_mean + k0, T, path);
        int leader_start = -1;
        int t = 0;
        while (t < T) {
            int e = t;
            while (e + 1 < T && path[e + 1] == path[t])
                e++;
            const int st = path[t];
            if (st != A && st != LH && st != LL) {
                leader_start = -1;
            } else {
                if (leader_start < 0)
                    leader_start = t;
                if (st == A) {
                    const int64_t ev_last = k0 + e;
                    const int64_t adapter_end = ev_last == n_events - 1 ? starts[ev_last] + 1 : starts[ev_last + 1];
                    const int64_t leader_in_read = starts[k0 + leader_start];
                    const int64_t total_duration = adapter_end - leader_in_read;
                    const int64_t adapter_duration = adapter_end - starts[k0 + t];
                    const int strict = (leader_in_read - payload_start) <= strict_duration;
                    if (total_duration >= cut_total[strict] && adapter_duration >= cut_adapter[strict]) {
                        if (count < cap) {
                            intervals[2 * count] = leader_in_read;
                            intervals[2 * count + 1] = 1 + adapter_end;
                        }
                        count++;
                    }
                    leader_start = -1;
                }
            }
            t = e + 1;
        }
    }
    free(path);
    return count;
}
