/*
 * pxo_pipeline.c -- ORACLE (test infrastructure, never shipped): the numeric
 * part of SignalAnalyzer.process / SignalAnalysis.process for one read and a
 * batch driver (the thing bench.py times as cpu_baseline, kind "port").
 * Stage order and early exits follow signal_analyzer.py:82-134,230-286.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "pxo.h"

static void result_init(pxg_read_result* r)
{
    memset(r, 0, sizeof(*r));
    r->status = PXG_ST_OKAY;
    for (int s = 0; s < PXG_N_SEGMENTS; s++)
        r->seg_first[s] = r->seg_last[s] = -1;
    r->bc_label = -1;
}

void pxo_process_read(const pxg_config* cfg, const int16_t* raw, int64_t n_raw,
                      const pxg_calib* cal, const float* ss_inject,
                      uint32_t stage_mask, pxg_read_result* r, pxg_polya_spike* spikes,
                      int spike_cap)
{
    const int stride = cfg->stride;
    result_init(r);
    r->n_pooled = (int32_t)(n_raw / stride);

    /* PHASE 1+2: prepare_loading + fit_scalers (signal_loader.py:77-109) */
    float ss[2] = { 1.0f, 0.0f };
    if (stage_mask & PXG_STAGE_SCALER) {
        const int n_head = cfg->scaler_length / stride;
        float* head = (float*)malloc(sizeof(float) * n_head);
        int st = pxo_head_pool(raw, n_raw, cal, cfg->scaler_length, stride,
                               cfg->scaler_min_length, head);
        if (st != PXG_ST_OKAY) {
            free(head);
            r->status = st;
            return;
        }
        if (ss_inject) {
            ss[0] = ss_inject[0];
            ss[1] = ss_inject[1];
        } else {
            pxo_scaler_forward(cfg, head, n_head, r->scaler_pred);
            st = pxo_scaler_transform(cfg, r->scaler_pred, ss);
        }
        free(head);
        r->scale = ss[0];
        r->shift = ss[1];
        if (st != PXG_ST_OKAY) {
            r->status = st;
            return;
        }
    } else if (ss_inject) {
        ss[0] = ss_inject[0];
        ss[1] = ss_inject[1];
        r->scale = ss[0];
        r->shift = ss[1];
    }
    if (!(stage_mask & PXG_STAGE_SEGMENT))
        return;

    /* PHASE 3: SignalAnalysis.process (signal_analyzer.py:230-286) */
    const int64_t P = n_raw / stride;
    float* pooled = (float*)malloc(sizeof(float) * (P > 0 ? P : 1));
    pxo_pool_scale(raw, n_raw, cal, stride, ss[0], ss[1], pooled);
    int T = (int)P;
    const int scan = cfg->segmentation_scan_limit / stride;       /* :347-349 */
    if (T > scan)
        T = scan;
    int32_t* path = (int32_t*)malloc(sizeof(int32_t) * (T > 0 ? T : 1));
    pxo_viterbi(&cfg->segmentation_model, pooled, T, path);
    pxo_segments(path, T, r->seg_first, r->seg_last);
    free(path);
    const int A = cfg->segmentation_model.adapter_state;
    if (A < 0 || r->seg_first[A] < 0) {                           /* :239-240 */
        r->status = PXG_ST_ADAPTER_NOT_DETECTED;
        free(pooled);
        return;
    }
    if (stage_mask & PXG_STAGE_BARCODE) {                         /* :247-248 */
        const int a0 = r->seg_first[A], a1 = r->seg_last[A];
        float* win = (float*)malloc(sizeof(float) * cfg->signal_trim_length);
        const int pushed = pxo_barcode_window(cfg, pooled + a0, a1 - a0 + 1, win);
        r->bc_pushed = (int8_t)pushed;
        if (pushed) {
            float probs[PXG_MAX_CLASSES];
            pxo_demux_forward(cfg, win, cfg->signal_trim_length, probs);
            pxo_barcode_call(cfg, probs, r);
        }
        free(win);
    }
    if (stage_mask & PXG_STAGE_POLYA) {                           /* :251-256 */
        const int PA = cfg->segmentation_model.polya_state;
        float* full = (float*)malloc(sizeof(float) * (n_raw > 0 ? n_raw : 1));
        pxo_raw_to_pa(raw, n_raw, cal, full);
        for (int64_t i = 0; i < n_raw; i++) {
            float y = ss[0] * full[i];
            full[i] = y + ss[1];
        }
        if (PA >= 0 && r->seg_first[PA] >= 0)
            pxo_polya(cfg, full, n_raw, r->seg_first[PA], r->seg_last[PA],
                      cal->sampling_rate, r, spikes, spike_cap);
        else
            pxo_polya(cfg, full, n_raw, r->seg_last[A] + 1, -1, cal->sampling_rate, r, spikes, spike_cap);
        free(full);
    }
    free(pooled);
}

void pxo_process_batch(const pxg_config* cfg, int64_t n_reads, const int16_t* raw_arena,
                       const int64_t* raw_offsets, const pxg_calib* calib,
                       const float* ss_inject, uint32_t stage_mask, pxg_read_result* out,
                       pxg_polya_spike* spikes, int spike_cap)
{
    for (int64_t i = 0; i < n_reads; i++)
        pxo_process_read(cfg, raw_arena + raw_offsets[i], raw_offsets[i + 1] - raw_offsets[i],
                         &calib[i], ss_inject ? ss_inject + 2 * i : NULL, stage_mask, &out[i],
                         spikes ? spikes + (size_t)i * spike_cap : NULL, spike_cap);
}
