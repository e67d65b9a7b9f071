/*
 * pxo.h -- ORACLE: CPU restatement of poreplex's raw-signal hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product (poreplex_amd/, libpxg.so) never links, imports or calls it.
 *
 * Every function restates one row of SURVEY.md section 8(a) and cites the
 * reference file:line it follows.  Struct layouts are taken from the ABI
 * contract header include/pxg.h (types only) so results compare field by
 * field with the HIP path.
 *
 * PARITY PINNING (see DESIGN.md "Oracle"):
 *   pinned by reference outputs generated in the build container
 *     (tools/make_golden.py, /opt/conda/bin/python3.9, real reference glue +
 *     compiled reference C): a1 a2 a5 a8 a9 a10 a11 a13 a14 a15 a16 a17 and the
 *     scaler de-standardise/QC arithmetic of a4;
 *   PARITY UNPINNED (TensorFlow and pomegranate are not installable here and
 *     the reference ships no tests): the LSTM forward passes (a4, a12) and the
 *     HMM Viterbi (a6, a7) are restated from Keras 2.2.4 / pomegranate>=0.10
 *     documented semantics and cross-checked against independent float64 /
 *     torch-CPU implementations and exhaustive path enumeration.
 */
#ifndef PXO_H
#define PXO_H

#include <stdint.h>
#include "../include/pxg.h"

#ifdef __cplusplus
extern "C" {
#endif

/* a1  fast5_file.py:122-131 */
void pxo_raw_to_pa(const int16_t* raw, int64_t n, const pxg_calib* cal, float* out);

/* numpy float32 add.reduce over a contiguous run (pairwise, 8 accumulators) */
float pxo_np_sum_f32(const float* a, int64_t n);

/* a2  signal_loader.py:212-231; returns 0 or PXG_ST_SCALER_SIGNAL_TOO_SHORT */
int pxo_head_pool(const int16_t* raw, int64_t n_raw, const pxg_calib* cal,
                  int length_limit, int stride, int min_length, float* out);

/* a5  signal_loader.py:233-264 (pool=stride, scale=True); returns P */
int64_t pxo_pool_scale(const int16_t* raw, int64_t n_raw, const pxg_calib* cal,
                       int stride, float scale, float shift, float* out);

/* canonical float32 transcendental kit shared (by specification, not by code)
 * with the HIP kernels: DESIGN.md "Canonical LSTM arithmetic" */
float pxo_expf(float x);
float pxo_sigmoid(float x);
float pxo_tanh(float x);
/* the 1024 x 4 spline coefficients both sides must agree on */
void pxo_sigmoid_table(float* out);

/* a4  signal_loader.py:96-97 + scaler-r3.hdf5; x[T] -> pred[2] */
void pxo_scaler_forward(const pxg_config* cfg, const float* x, int T, float* pred);
/* a4  signal_loader.py:98-109; returns 0 or PXG_ST_SCALING_QC_FAIL */
int pxo_scaler_transform(const pxg_config* cfg, const float* pred, float* scale_shift);

/* a12 barcoding.py:106-107 + demux-tetra-r4.hdf5; x[T] -> probs[n_classes] */
void pxo_demux_forward(const pxg_config* cfg, const float* x, int T, float* probs);
/* intermediate access for layer-by-layer tests: runs one LSTM layer over a
 * sequence; seq_out may be NULL; reverse!=0 consumes x back to front and
 * stores outputs at their original time index (Keras Bidirectional). */
void pxo_lstm_layer(const pxg_lstm_layer* L, const float* x, int T, int reverse,
                    float* seq_out, float* h_last);

/* the same layer in the exact fixed-point arithmetic (PXG_LSTM_Q8, see pxo_core.c): scalar inputs
 * as floats xs[T] (input_dim 1) or vector inputs as q values xq[T][input_dim] (h = q * 2^-22);
 * outputs are q values */
void pxo_lstm_layer_q(const pxg_lstm_layer* L, const float* xs, const int32_t* xq, int T, int reverse,
                      int32_t* seq_out, int32_t* q_last);

/* a6+a7 worker_persistence.py:95-121 + pomegranate viterbi; returns logp */
double pxo_viterbi(const pxg_hmm* hmm, const float* x, int T, int32_t* path);
/* log emission density of state s at x (pomegranate Normal / GMM) */
double pxo_hmm_emission(const pxg_hmm* hmm, int s, double x);
/* a8  signal_analyzer.py:346-364 (run-length summary, last run wins) */
void pxo_segments(const int32_t* path, int T, int32_t* seg_first, int32_t* seg_last);

/* a11 barcoding.py:77-81 (in place on a copy) */
void pxo_normalize_signal(const float* sig, int n, float* out);
/* a9+a10 signal_analyzer.py:445-448, barcoding.py:83-101; returns pushed flag */
int pxo_barcode_window(const pxg_config* cfg, const float* adapter_signal, int len,
                       float* out);
/* a13 barcoding.py:72-75 */
int pxo_phred(const pxg_config* cfg, float score);
/* a12 tail: barcoding.py:108-118 */
void pxo_barcode_call(const pxg_config* cfg, const float* probs, pxg_read_result* r);

/* a15 src/csupport.c:70-124 -> scrappie event_detection.c; returns n events */
int64_t pxo_detect_events(const float* sig, int64_t n, int64_t w1, int64_t w2,
                          float thr1, float thr2, float peak_height,
                          pxg_event* out, int64_t max_out);
/* scipy.signal.medfilt(x, k) for odd k, zero padded (polya.py:62-63) */
void pxo_medfilt(const float* x, int64_t n, int k, float* out);

/* a14+a16+a17 polya.py:50-187 on one read; fills polya_* fields of r */
void pxo_polya(const pxg_config* cfg, const float* scaled_full, int64_t n_raw,
               int rough_begin, int rough_end_or_neg, double sampling_rate,
               pxg_read_result* r, pxg_polya_spike* spikes_or_null, int spike_cap);
/* (every spike is counted in r->polya_n_spikes; the first spike_cap rows are stored) */
/* a16 polya.py:156-187; is_polya/length per event; returns 1 and (i,j) or 0 */
int pxo_best_polya_interval(const pxg_config* cfg, const uint8_t* is_polya,
                            const float* length, int n_events, int* out_i, int* out_j);

/* a18 fast5_file.py:210-230 (Guppy): medfilt(5) of pA[first, first+stride*n) ->
 * block means (float32 pairwise), NaN where the read ends early; scaled =
 * fl(fl(scale*mean)+shift) (signal_analyzer.py:318).  Returns 0, or -1 when the
 * reference would raise 'Numbers of events and raw data strides does not match' */
int pxo_guppy_event_means(const int16_t* raw, int64_t n_raw, const pxg_calib* cal,
                          int64_t first_sample, int64_t n_events, int stride, float scale,
                          float shift, float* mean, float* scaled);
int pxo_guppy_event_table(const int16_t* raw, int64_t n_raw, const pxg_calib* cal,
                          int64_t first, int64_t n_events, int stride, float scale,
                          float shift, float* mean, float* stdv_or_null, float* scaled);
/* a19 signal_analyzer.py:366-418: window scan -> candidate [leader start,
 * adapter end + 1] intervals; returns their number (stores at most cap) */
int pxo_unsplit_scan(const pxg_config* cfg, const float* scaled_mean, int64_t n_events,
                     int64_t first_sample, int stride, int64_t payload_start,
                     double sampling_rate, int64_t* intervals, int cap);

/* a19 over a table with its own events (albacore Events): starts[] ascending, see pxo_unsplit.c */
int pxo_unsplit_scan_events(const pxg_config* cfg, const float* scaled_mean, const int64_t* starts,
                            int64_t n_events, int64_t payload_start, double rate, int64_t* intervals, int cap);

/* whole per-read path (signal_analyzer.py:82-134 phases 1-4, numeric part) */
void pxo_process_read(const pxg_config* cfg, const int16_t* raw, int64_t n_raw,
                      const pxg_calib* cal, const float* scale_shift_or_null,
                      uint32_t stage_mask, pxg_read_result* out,
                      pxg_polya_spike* spikes_or_null, int spike_cap);
void pxo_process_batch(const pxg_config* cfg, int64_t n_reads, const int16_t* raw_arena,
                       const int64_t* raw_offsets, const pxg_calib* calib,
                       const float* scale_shift_or_null, uint32_t stage_mask,
                       pxg_read_result* out, pxg_polya_spike* spikes_or_null /* n x spike_cap */,
                       int spike_cap);

#ifdef __cplusplus
}
#endif
#endif
