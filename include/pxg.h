/*
 * pxg.h -- C ABI of the MI355X-native poreplex raw-signal hot path (libpxg.so).
 *
 * The reference (hyeshik/poreplex) has no FFI for this path: the per-read
 * processor is Python (poreplex/signal_analyzer.py:46-134) calling NumPy,
 * TensorFlow/Keras, pomegranate and one in-tree C extension
 * (src/csupport.c:70-124).  This header is the boundary a maintainer would
 * bind with ctypes (INTEGRATION.md shows the stub) so that
 * `process_batch(batchid, reads, config)` keeps its signature while every
 * stage between "int16 DAQ samples" and "per-read result record" runs on the
 * GPU.  Each entry point cites the reference code it replaces.
 *
 * Conventions (mirroring src/csupport.c:92-121): caller owns every host
 * buffer; nothing is retained after return (one exception: the arrays handed
 * to pxg_batch_stage are read until pxg_batch_swap returns); functions return 0 or a negative
 * pxg_error and never throw; per-read domain failures are DATA (the `status`
 * field), not errors.  A context is bound to one GPU.  pxg_process_batch / pxg_process_batch_ex may be
 * called from several host threads at once (the calls overlap on the device); every other entry point
 * belongs to ONE host thread at a time.
 * Plain pointers and sizes only -- no torch / HIP types in any signature.
 */
#ifndef PXG_H
#define PXG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PXG_ABI_VERSION 5

#define PXG_MAX_STATES      8   /* HMM states per model (reference uses 6)        */
#define PXG_MAX_MIXTURE     4   /* Gaussian components per state (reference <= 2) */
#define PXG_N_SEGMENTS      8   /* seg_first/seg_last slots (= PXG_MAX_STATES)     */
#define PXG_MAX_CLASSES     8   /* softmax width (reference: 1 decoy + 4 barcodes) */
#define PXG_MAX_CALIBRATION 64  /* phred calibration table rows (reference: 29)    */

/* per-read error code pxg_batch_unsplit_scan stores in out_count (data problems are
 * per-read results, never call failures -- signal_analyzer.py:118-122) */
#define PXG_UNSPLIT_E_GEOMETRY     (-3) /* negative first_sample / n_blocks                  */
#define PXG_UNSPLIT_E_WINDOW       (-4) /* a scan window outgrew the slots the host sized for it (a bound of
                                           the library proved wrong): the read fails loudly, no candidate is dropped */

/* ---- error codes (function return values) -------------------------------- */
enum pxg_error {
    PXG_OK = 0,
    PXG_E_INVALID = -1,     /* bad argument / inconsistent config          */
    PXG_E_NODEVICE = -2,    /* no usable HIP device                        */
    PXG_E_HIP = -3,         /* a HIP runtime call failed (see last_error)  */
    PXG_E_NOMEM = -4,
    PXG_E_STATE = -5,       /* call order violated (e.g. run before upload) */
    PXG_E_UNSUPPORTED = -6
};

/* ---- per-read status: the reference's status strings ----- *
 * signal_loader.py:123,206,221,109; signal_analyzer.py:90-92,240,273,279,342 */
enum pxg_status {
    PXG_ST_OKAY = 0,
    PXG_ST_DISAPPEARED = 1,
    PXG_ST_IRREGULAR_FAST5 = 2,
    PXG_ST_SCALER_SIGNAL_TOO_SHORT = 3,
    PXG_ST_SCALING_QC_FAIL = 4,
    PXG_ST_ADAPTER_NOT_DETECTED = 5,
    PXG_ST_NOT_BASECALLED = 6,
    PXG_ST_BASECALL_TABLE_INCOMPLETE = 7,
    PXG_ST_UNSPLIT_READ = 8,
    PXG_ST_SEQUENCE_TOO_SHORT = 9,
    PXG_ST_UNKNOWN_ERROR = 10,
    PXG_N_STATUS = 11
};

/* ---- stage selection bits for pxg_process_batch / pxg_batch_run ---------- */
enum pxg_stage {
    PXG_STAGE_SCALER   = 1u << 0, /* a2-a4: head pool + scaler LSTM + QC        */
    PXG_STAGE_SEGMENT  = 1u << 1, /* a1,a5,a7,a8: pA + pool + scale + Viterbi   */
    PXG_STAGE_BARCODE  = 1u << 2, /* a9-a13: window, robust z, demux LSTM       */
    PXG_STAGE_POLYA    = 1u << 3, /* a14-a17: event detection + interval DP     */
    PXG_STAGE_ALL_DEMUX = (1u << 0) | (1u << 1) | (1u << 2)
};

/* ---- arithmetic of the recurrent matrix products of a4 / a12 (pxg_config.lstm_arith) ----
 * The reference hands both networks to TensorFlow (signal_loader.py:96-97, barcoding.py:106-107):
 * float32 Keras LSTM equations, summation order unspecified.  Two canonical evaluations exist
 * here, each restated bit for bit by oracle/pxo_core.c and selectable per context:
 *   PXG_LSTM_Q8   hidden states and weights as 3-digit balanced base-256 fixed point (h: 2^-22,
 *                 weights: 2^-p per matrix), the six leading digit products summed EXACTLY on the
 *                 int8 matrix pipe (v_mfma_i32_16x16x64_i8), one float32 rounding per gate;
 *                 integer sums are order-independent, so the result does not depend on how the
 *                 hardware adds.  Default.
 *   PXG_LSTM_F32  one k-ordered float32 fma chain per gate on v_mfma_f32_16x16x4_f32 (rounds 1-3). */
enum pxg_lstm_arith { PXG_LSTM_Q8 = 0, PXG_LSTM_F32 = 1 };

/* indices into pxg_stage_times.ms[] */
enum pxg_timer {
    PXG_T_HEAD_POOL = 0, PXG_T_SCALER_LSTM, PXG_T_SEGMENT, PXG_T_BARCODE_WINDOW,
    PXG_T_DEMUX_BIDIR, PXG_T_DEMUX_TOP, PXG_T_POLYA, PXG_T_EVENT_MEANS, PXG_T_UNSPLIT,
    PXG_T_FINALIZE, PXG_T_TOTAL,
    PXG_N_TIMERS
};

/* DAQ calibration of one read: fast5_file.py:110-115 (channel_id attrs). */
typedef struct {
    double range;
    double digitisation;
    double offset;
    double sampling_rate;
} pxg_calib;

/* One HMM (worker_persistence.py:95-121 + rna-r941.cfg:61-151), states in the
 * order the preset lists them.  trans[i][j] = P(i -> j); 0 means "no edge". */
typedef struct {
    int32_t n_states;
    int32_t adapter_state;                 /* index of 'adapter', -1 if none   */
    int32_t polya_state;                   /* index of 'polya-tail', -1 if none*/
    int32_t leader_low_state;              /* index of 'leader-low', -1 if none */
    int32_t leader_high_state;             /* index of 'leader-high', -1 if none*/
    int32_t reserved;
    int32_t name_rank[PXG_MAX_STATES];     /* rank of the state name in sorted
                                              order (pomegranate tie order)    */
    int32_t n_mix[PXG_MAX_STATES];
    double start_prob[PXG_MAX_STATES];
    double mix_mu[PXG_MAX_STATES][PXG_MAX_MIXTURE];
    double mix_sigma[PXG_MAX_STATES][PXG_MAX_MIXTURE];
    double mix_weight[PXG_MAX_STATES][PXG_MAX_MIXTURE];
    double trans[PXG_MAX_STATES][PXG_MAX_STATES];
} pxg_hmm;

/* Keras LSTM layer, Keras layout: kernel [in, 4*units], recurrent
 * [units, 4*units], bias [4*units]; gate blocks ordered i, f, c, o. */
typedef struct {
    int32_t input_dim;
    int32_t units;
    const float* kernel;
    const float* recurrent;
    const float* bias;
} pxg_lstm_layer;

typedef struct {
    int32_t in_dim;
    int32_t out_dim;
    const float* kernel; /* [in_dim, out_dim] */
    const float* bias;   /* [out_dim] */
} pxg_dense_layer;

/* Flattened numeric image of the reference's config dict (keys
 * read at signal_analyzer.py:231-278,347; signal_loader.py:50-65,93;
 * barcoding.py:52-53,84-90,107; polya.py:32-48). */
typedef struct {
    uint32_t abi_version;       /* must be PXG_ABI_VERSION                       */
    int32_t device_id;          /* HIP device ordinal (rank -> GPU)              */

    /* signal_processing (rna-r941.cfg:8-12) + scaler-r3 attrs */
    int32_t stride;             /* rough_signal_stride = 15                      */
    int32_t scaler_length;      /* input_defs.length = 30000                     */
    int32_t scaler_min_length;  /* input_defs.min_length = 9000                  */
    int32_t lstm_arith;         /* enum pxg_lstm_arith: arithmetic of the recurrent matmuls (a4, a12) */
    double scaler_xfrm[4];      /* scale_mean, scale_std, shift_mean, shift_std  */
    double scaler_qc_scale[2];  /* inclusive bounds, norm.ppf (signal_loader.py:65-68) */
    double scaler_qc_shift[2];
    pxg_lstm_layer scaler_lstm1, scaler_lstm2;
    pxg_dense_layer scaler_dense;

    /* segmentation (rna-r941.cfg:14-15,61-101) */
    int32_t segmentation_scan_limit;  /* raw samples, 100000 */
    int32_t reserved1;
    pxg_hmm segmentation_model;
    pxg_hmm unsplit_model;            /* carried for the chimera filter (a19)    */

    /* unsplit_read_detection (rna-r941.cfg:17-27), seconds / fractions */
    double unsplit_window_size, unsplit_window_step, unsplit_strict_duration;
    double unsplit_strict_full_length, unsplit_strict_dna_length;
    double unsplit_loosen_full_length, unsplit_loosen_dna_length;

    /* demultiplexing (rna-r941.cfg:29-36) */
    int32_t number_of_decoy_labels;
    int32_t number_of_barcodes;
    int32_t minimum_dna_length;
    int32_t maximum_dna_length;
    int32_t signal_trim_length;
    int32_t n_calibration;
    double calibration[PXG_MAX_CALIBRATION];
    double score_threshold;           /* calibration[barcoding_quality_filter]   */
    float pad_filler;                 /* barcoding.py:32  (-1000.0)              */
    int32_t reserved2;
    pxg_lstm_layer demux_fwd, demux_bwd, demux_top;
    pxg_dense_layer demux_dense;

    /* polya_dwell (rna-r941.cfg:38-59) */
    int32_t polya_refinement_expansion;
    int32_t polya_openend_expansion;
    int32_t polya_median_pre_filter;
    int32_t polya_maximum_openend_extension;
    int32_t ed_window_length1, ed_window_length2;
    float ed_threshold1, ed_threshold2, ed_peak_height;
    int32_t polya_spike_tolerance;
    double polya_mean_dist[2];
    double polya_mean_z_cutoff;
    double polya_stdv_max;
    double polya_stdv_range[2];
    double polya_spike_weight;
    double polya_mean_trigger_recalibration;
    int32_t recal_max_dist_from_adapter;
    int32_t recal_min_length;
    double recal_max_stdv;
} pxg_config;

/* Per-read result record: what NanoporeRead carries out of the hot path
 * (signal_loader.py:114-155,165-198).  POD, 8-byte aligned, also the unit of
 * the multi-GPU all-gather. */
typedef struct {
    int32_t status;                       /* enum pxg_status                     */
    int32_t n_pooled;                     /* P = N // stride                     */
    int32_t seg_first[PXG_N_SEGMENTS];    /* pooled, right-inclusive; -1 absent  */
    int32_t seg_last[PXG_N_SEGMENTS];
    float scale, shift;                   /* scaling_params (float32)            */
    float scaler_pred[2];                 /* raw network output                  */
    int8_t bc_pushed;                     /* adapter window passed the length gate */
    int8_t bc_called;                     /* barcode != None                     */
    int8_t bc_label;                      /* argmax - n_decoy (barcode_guess)    */
    uint8_t bc_phred;                     /* calibrated score                    */
    float bc_score;                       /* max softmax                         */
    float probs[PXG_MAX_CLASSES];
    int8_t polya_called;
    int8_t reserved8;
    int16_t reserved16;
    int32_t polya_n_spikes;               /* every spike event of the tail (polya.py:109-115); the
                                             rows themselves: pxg_batch_download_spikes          */
    int32_t polya_dwell_samples;          /* sum of poly(A)-event lengths        */
    int32_t reserved32;                   /* 0 (no implicit padding: records compare byte for byte) */
    int64_t polya_begin, polya_end;       /* raw-sample coordinates              */
} pxg_read_result;

/* poly(A) spike detail, kept OUT of the gathered record: one row per spike,
 * (length, mean of the event before, of the spike, of the event after) --
 * polya.py:111-114.  As many rows per read as the tail has spikes (the reference lists
 * all of them): rows of all reads back to back in read order + offsets (CSR). */
typedef struct {
    float v[4];
} pxg_polya_spike;

/* scrappie event record (src/contrib/scrappie/scrappie_structures.h:8-15,
 * as exported by src/csupport.c:156-159). */
typedef struct {
    uint64_t start;
    float length;
    float mean;
    float stdv;
    int32_t pos;
    int32_t state;
} pxg_event;

typedef struct {
    float ms[PXG_N_TIMERS];      /* HIP-event time of each stage, last run       */
    int64_t n_launches[PXG_N_TIMERS];
} pxg_stage_times;

typedef struct {
    char name[128];
    char arch[32];
    int32_t compute_units;
    int32_t wavefront_size;
    int64_t total_mem;
    int32_t lds_per_cu;
    int32_t clock_khz;
} pxg_device_info;

typedef struct pxg_ctx pxg_ctx;
typedef struct pxg_z_chunk pxg_z_chunk;   /* defined with the sample codec below */

/* ---- lifetime: replaces WorkerPersistenceStorage.init_persistence_objects
 * (worker_persistence.py:60-90): load models once per worker process. ------ */
int pxg_create(const pxg_config* cfg, pxg_ctx** out);
void pxg_destroy(pxg_ctx* ctx);
const char* pxg_last_error(const pxg_ctx* ctx_or_null);
int pxg_abi_version(void);
int pxg_get_device_info(pxg_ctx* ctx, pxg_device_info* out);
/* PCI address ("0000:c1:00.0") of HIP device `device`: what a launcher needs to put a rank's
 * process and its page-locked staging memory on the NUMA node of ITS GPU (one worker process per
 * device: pipeline.py:96).  No context needed. */
int pxg_device_pci_bus_id(int device, char* out, int cap);

/* ---- the hot path: replaces SignalAnalyzer.process for the numeric part
 * (signal_analyzer.py:82-134 phases 1-4).  raw_offsets has n_reads+1 entries
 * (sample indices into raw_arena).  scale_shift_or_null: n_reads x 2 float32
 * to INJECT scaling_params and skip the scaler network (test hook). --------- */
int pxg_process_batch(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                      const int64_t* raw_offsets, const pxg_calib* calib,
                      const float* scale_shift_or_null, uint32_t stage_mask,
                      pxg_read_result* out);

/* The same call with everything a worker batch may need around the records, and the form
 * the reference's `parallel` workers map onto (pipeline.py:96,204-205): SEVERAL HOST THREADS
 * MAY CALL pxg_process_batch / pxg_process_batch_ex ON ONE CONTEXT AT THE SAME TIME.  Calls
 * overlap on the device: a call copies its samples into the spare input slot on the copy
 * stream while the previous call's kernels run on the resident batch, becomes resident when
 * that call has its results, runs every stage, and returns with everything on the host --
 * the copy of call k+1 under the kernels of call k, the caller's host work for call k-1
 * (result dicts) under both.  Inputs: page-lock them (pxg_host_register) for DMA transfers.
 * (The split calls pxg_batch_stage/swap/run/... may run beside it on other threads under pxg_ctx_lock.)
 *   z != NULL: the samples arrive encoded (pxg_batch_stage_z's arguments), raw_arena is NULL.
 *   unsplit_first_sample != NULL: also run the a19 window scan (pxg_batch_unsplit_scan's
 *     arguments) on the batch while it is resident; candidates of all reads back to back in
 *     unsplit_intervals (cap pairs), unsplit_count per read, unsplit_total pairs found.
 *   spikes != NULL (poly(A) stage): the spike rows of all reads in read order, spike_offsets
 *     n_reads + 1, spike_total rows found.
 * A variable-size output that did not fit its cap leaves its *_total set and the call returns
 * PXG_E_NOMEM: call again with buffers of that size (records and counts are valid). */
typedef struct {
    uint32_t struct_bytes;                 /* sizeof(pxg_batch_extras): layout guard            */
    int32_t unsplit_block_stride;
    const float* scale_shift_or_null;
    const uint8_t* z; int64_t z_bytes; const pxg_z_chunk* chunks; int64_t n_chunks;
    int64_t data_base, dst_base;
    const int64_t* unsplit_first_sample; const int64_t* unsplit_n_blocks;
    int64_t unsplit_cap; int64_t* unsplit_intervals; int32_t* unsplit_count; int64_t unsplit_total;
    int64_t spike_cap; pxg_polya_spike* spikes; int64_t* spike_offsets; int64_t spike_total;
} pxg_batch_extras;
int pxg_process_batch_ex(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena_or_null,
                         const int64_t* raw_offsets, const pxg_calib* calib, uint32_t stage_mask,
                         pxg_batch_extras* extras_or_null, pxg_read_result* out);
/* SMALL CALLS SHARE A BATCH.  The reference hands a worker 128 reads at a time (commandline.py:402 `--batch-size`),
 * and 128 reads cost the kernels what 4 000 cost (their latency floor, ~5.5 ms).  Calls of at most 4 096 reads with
 * plain inputs (int16 samples; no injected scaling, encoded samples, window scan or spike rows) that arrive from
 * several threads while the pipeline is full -- one batch computing, one staged -- are gathered and run as ONE
 * batch: the first caller of a group waits for the spare slot on everybody's behalf, copies each call's samples
 * from where they lie, runs the stages and hands every call its slice of the records.  A lone call starts at once;
 * groups grow with the load, up to 65 536 reads (a call that would overfill a group runs alone behind it).  Records
 * are those of separate calls (each is a function of its read alone).  What a merged call does NOT keep to itself: an
 * error of the shared batch (out of memory, a HIP error) is returned to every call of the group with a message that
 * starts "merged", and pxg_batch_times / the resident-batch queries after such a call describe the whole group.
 * pxg_merge_stats: groups run so far and the calls they carried (PXG_NO_CALL_MERGE=1 at pxg_create switches it off). */
int pxg_merge_stats(pxg_ctx* ctx, int64_t* groups, int64_t* calls);

/* Split form for device-resident batches (loader overlap, benchmarking):
 * upload = H2D copy into context-owned HBM arenas; run = enqueue all kernels
 * on the context stream (asynchronous); sync = wait; download = D2H results. */
int pxg_batch_upload(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                     const int64_t* raw_offsets, const pxg_calib* calib,
                     const float* scale_shift_or_null);
/* Double-buffered input for a loader (phase 1 of SignalAnalyzer.process,
 * signal_analyzer.py:82-100, overlapped with the numeric phases of the previous
 * batch): pxg_batch_stage copies the NEXT batch into the spare input slot on a
 * dedicated copy stream while the resident batch computes; pxg_batch_swap waits
 * for those copies and makes the staged batch the resident one (download the
 * previous batch's results first).  The host arrays must stay valid until
 * pxg_batch_swap returns; pin them (pxg_host_register) for true DMA transfers. */
int pxg_batch_stage(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                    const int64_t* raw_offsets, const pxg_calib* calib,
                    const float* scale_shift_or_null);
/* Large synthetic batches without staging them on the host (configs[3] is
 * 12 GB of int16, configs[4] 15 GB per GPU): upload `base_n` distinct reads once and
 * make the resident batch read j = base read (phase + j) % base_n, replicated on the
 * device.  Every stage then runs on n_reads independent reads exactly as after
 * pxg_batch_upload.  phase lets rank r of a sharded run own reads [lo, hi) of ONE
 * global tiling (phase = lo % base_n). */
int pxg_batch_upload_tiled(pxg_ctx* ctx, int64_t n_reads, int64_t base_n, int64_t phase,
                           const int16_t* base_arena, const int64_t* base_offsets,
                           const pxg_calib* base_calib, const float* base_scale_shift_or_null);
int pxg_batch_swap(pxg_ctx* ctx);
/* Page-lock / release a host array for DMA transfers.  Optional: a host -> device copy of 512 KB or more from an
 * array that is NOT page-locked travels through page-locked chunks of the context (the GPU never addresses pageable
 * memory of the caller; the runtime's own in-place lock of such a source is what took the process down in round 4:
 * INTEGRATION.md section 2).  Register MAPPED pages of their own (mmap), never a slice of a malloc'ed block: a
 * brk-heap range that was page-locked once must not be page-locked again later in the process.
 * pxg_host_unregister accepts ctx == NULL: page locks are process-wide, and a range may have to be released after the
 * context that registered it is gone (a staging arena must not be unmapped while it is still registered). */
int pxg_host_register(pxg_ctx* ctx, void* ptr, size_t bytes);
int pxg_host_unregister(pxg_ctx* ctx, void* ptr);
int pxg_batch_run(pxg_ctx* ctx, uint32_t stage_mask);
int pxg_batch_sync(pxg_ctx* ctx);
/* sync / download also finish the poly(A) stage: a read whose inspection window needed more
 * event rows than the first pass hands out (open-ended extension over a long featureless
 * signal) is re-run with what it asked for before any record leaves the device, so a
 * record never depends on the size of a scratch buffer. */
int pxg_batch_download(pxg_ctx* ctx, pxg_read_result* out);
/* spike rows of the last run (poly(A) stage only): offsets[r] .. offsets[r + 1] = the rows of
 * read r (offsets has n_reads + 1 entries and is always filled; offsets[n_reads] = the sum of the
 * records' polya_n_spikes).  PXG_E_NOMEM when cap_rows is smaller than that (no row written). */
int pxg_batch_download_spikes(pxg_ctx* ctx, int64_t cap_rows, pxg_polya_spike* out, int64_t* offsets);
/* The pooled + scaled signal (a5, load_signal(pool=stride) with the scale / shift of the last
 * run) of ONE stretch per read of the resident batch -- dump_adapter_signal's
 * signal[adapter_first : adapter_last + 1] (signal_analyzer.py:450-466): read r contributes the
 * pooled positions first[r] ... into out[out_offsets[r] : out_offsets[r + 1]] (n + 1 offsets from
 * 0; an empty stretch leaves a read out).  Only for reads the scaler stage succeeded on. */
int pxg_batch_pooled_signal(pxg_ctx* ctx, const int64_t* first, const int64_t* out_offsets, float* out);
/* The barcode windows of the resident batch as the classifier saw them (a9-a11: the adapter
 * stretch cut to / padded to signal_trim_length and normalised, barcoding.py:77-101), n x
 * signal_trim_length; only the rows of reads whose record says bc_pushed are meaningful.  What a
 * training set for the demultiplexer is made of (training/barcodes/scripts/prepare_training_data.py:
 * 63-87 prepares its inputs exactly as inference does). */
int pxg_batch_download_windows(pxg_ctx* ctx, float* out);
int pxg_batch_times(pxg_ctx* ctx, pxg_stage_times* out);

/* ---- per-stage hooks (parity tests call these through the same ABI) ------ */
/* a1: Fast5Reader.get_raw_data (fast5_file.py:122-131) */
int pxg_raw_to_pa(pxg_ctx* ctx, int64_t n, const int16_t* raw, const pxg_calib* calib,
                  float* out);
/* a2: NanoporeRead.load_padded_signal_head (signal_loader.py:212-231);
 * out: n_reads x (scaler_length/stride) float32; status: n_reads int32 */
int pxg_head_pool(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                  const int64_t* raw_offsets, const pxg_calib* calib, float* out,
                  int32_t* status);
/* a4: scaler_model.predict (signal_loader.py:96-97); head n x T, pred n x 2 */
int pxg_scaler_lstm(pxg_ctx* ctx, int64_t n_reads, const float* head, float* pred);
/* a4: de-standardise + QC (signal_loader.py:98-109); status 0 or SCALING_QC_FAIL */
int pxg_scaler_transform(pxg_ctx* ctx, int64_t n_reads, const float* pred,
                         float* scale_shift, int32_t* status);
/* a5: NanoporeRead.load_signal(pool=stride) (signal_loader.py:233-264);
 * pooled_offsets n+1 (in pooled samples) describes the caller's out arena */
int pxg_pool_scale(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                   const int64_t* raw_offsets, const pxg_calib* calib,
                   const float* scale_shift, const int64_t* pooled_offsets, float* out);
/* a7+a8: segmodel.viterbi + detect_segments (signal_analyzer.py:346-364) on
 * already pooled+scaled signals; path_or_null receives state ids per step */
int pxg_viterbi(pxg_ctx* ctx, int which_model, int64_t n_reads, const float* signal_arena,
                const int64_t* signal_offsets, int32_t* seg_first, int32_t* seg_last,
                int32_t* path_or_null, double* logp_or_null);
/* a10+a11: BarcodeDemultiplexer.push/normalize_signal (barcoding.py:77-101);
 * out n x signal_trim_length, pushed n int8 */
int pxg_barcode_window(pxg_ctx* ctx, int64_t n_reads, const float* signal_arena,
                       const int64_t* signal_offsets, float* out, int8_t* pushed);
/* a12: demuxer.model.predict (barcoding.py:106-107); win n x T, probs n x classes */
int pxg_demux_lstm(pxg_ctx* ctx, int64_t n_reads, const float* win, float* probs);
/* a18: Guppy event table means (fast5_file.py:210-230: medfilt(5) of the pA
 * samples [first, first + 15*n_events), block means) and their scaled values
 * (signal_analyzer.py:318); events_offsets has n+1 entries into mean/scaled */
int pxg_guppy_event_means(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena,
                          const int64_t* raw_offsets, const pxg_calib* calib,
                          const float* scale_shift, const int64_t* first_sample,
                          const int64_t* events_offsets, int32_t block_stride,
                          float* mean, float* scaled_mean);
/* a18 on the RESIDENT batch with the column the event-table dump adds (--dump-basecalls,
 * signal_analyzer.py:183-190): mean, stdv (fast5_file.py:226-227) and scaled_mean
 * (signal_analyzer.py:318) of every Guppy block of every read; events_offsets (n + 1, from 0)
 * gives each read's blocks, none = the read is left out. */
int pxg_batch_event_table(pxg_ctx* ctx, const int64_t* first_sample, const int64_t* events_offsets,
                          int32_t block_stride, float* mean, float* stdv, float* scaled_mean);
/* a19 (numeric part): the window scan of detect_unsplit_read
 * (signal_analyzer.py:366-418) on the RESIDENT batch after a run with the
 * segment stage: per read first_sample_template and the number of Guppy
 * blocks.  Output is compact (CSR): out_count[r] >= 0 candidates of read r, or a
 * negative PXG_UNSPLIT_E_* code for that read alone; the (leader start, adapter
 * end + 1) pairs, raw-sample coordinates, of all reads back to back in read
 * order in out_intervals (cap_intervals x 2); *out_total = pairs found (when it
 * exceeds cap_intervals only the first cap_intervals were written: call again
 * with a larger buffer).  A window keeps as many candidates as its length and the duration
 * cut-offs of the config allow (the slots are sized from them): no fixed limit.  Everything between the H2D of the two per-read arrays
 * and the final D2H is enqueued without a host synchronisation.
 * Threading: the scan calls are stage hooks of the RESIDENT batch -- for their duration the context enqueues on its
 * scan stream (beside K6 when the last run had the poly(A) stage).  They must not overlap any other call on the same
 * context: a caller that shares the context between threads holds pxg_ctx_lock(ctx, 1) around run + scans + downloads
 * (pxg_process_batch_ex does). */
int pxg_batch_unsplit_scan(pxg_ctx* ctx, const int64_t* first_sample, const int64_t* n_blocks,
                           int32_t block_stride, int64_t cap_intervals, int64_t* out_intervals,
                           int32_t* out_count, int64_t* out_total);
/* a19 for reads whose basecall brings its OWN event table -- albacore's 14-column `Events', which
 * Fast5Reader.load_events returns unchanged (fast5_file.py:178-179): n_events[r] rows of read r (0 leaves
 * the read out, negative is that read's PXG_UNSPLIT_E_GEOMETRY), their `start' column (ascending, samples)
 * and float32 `mean' column back to back in read order.  scaled_mean = fl(fl(scale * mean) + shift) on the
 * device (signal_analyzer.py:318); a window holds the events with left <= start <= left + window_size
 * (:384-386); an event ends where the next starts, the last one sample after its start (:321-324).
 * Output and capacity handling as pxg_batch_unsplit_scan. */
int pxg_batch_unsplit_scan_events(pxg_ctx* ctx, const int64_t* n_events, const int64_t* ev_start,
                                  const float* ev_mean, int64_t cap_intervals, int64_t* out_intervals,
                                  int32_t* out_count, int64_t* out_total);
/* The two locks of pxg_process_batch_ex, for a caller that drives the split calls (stage / swap / run /
 * scans / downloads) from several threads and mixes them with the one-call form: which = 0 the spare input
 * slot (hold from pxg_batch_stage to pxg_batch_swap), 1 the resident batch (take before the swap, hold to
 * the last download); always 0 before 1. */
int pxg_ctx_lock(pxg_ctx* ctx, int which);
int pxg_ctx_unlock(pxg_ctx* ctx, int which);
/* a14-a17 as a standalone hook: PolyASignalAnalyzer.__call__ (polya.py:50-148) on
 * reads whose scaling and segmentation the caller supplies (seg_first/seg_last:
 * n x PXG_N_SEGMENTS, pooled right-inclusive, -1 absent, as pxg_viterbi returns
 * them).  out: n records of which only the polya_* fields (and status = OKAY) are
 * set; spike rows as pxg_batch_download_spikes returns them (spike_offsets_or_null n + 1
 * entries; PXG_E_NOMEM with the offsets filled when spike_cap rows are too few).  Settles a
 * resident batch's pending poly(A) retries first: safe between pxg_batch_run and its download. */
int pxg_polya(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena, const int64_t* raw_offsets,
              const pxg_calib* calib, const float* scale_shift, const int32_t* seg_first,
              const int32_t* seg_last, pxg_read_result* out, int64_t spike_cap,
              pxg_polya_spike* spikes_or_null, int64_t* spike_offsets_or_null);
/* a15: csupport.detect_events (src/csupport.c:70-124) on a batch of windows */
int pxg_detect_events(pxg_ctx* ctx, int64_t n_windows, const float* signal_arena,
                      const int64_t* signal_offsets, int64_t max_events_per_window,
                      pxg_event* events, int64_t* n_events);

/* ---- result sinks ------------------------------------------------------
 * The rows of sequencing_summary.txt (io.py:120-184, SequencingSummaryWriter.write_results)
 * as text, from columns -- byte for byte what the reference's print of str(int), repr(float),
 * round(start_time / sampling_rate, 3) and format(dwell, '.4f') produces.  Host only, exported by
 * libpxghost.so (no HIP runtime behind it, loads anywhere).  The
 * five string fields come straight from NumPy '<U' arrays (UCS-4, fixed width, NUL padded):
 * text[0..4] = filename, read_id, run_id, channel, sample_id, row string_row[k] of each.
 * Every other array has n elements.  barcode_names == NULL: barcoding off (no barcode /
 * score fields); else barcode_names[0] is the "no call" name and [b + 1] barcode b's, and
 * barcode[k] = -1 for no call.  polya_dwell == NULL: no poly(A) field.
 * Returns the number of bytes written to `out` (no terminator), PXG_E_NOMEM if `cap` is too
 * small, PXG_E_UNSUPPORTED for non-ASCII text or a start time beyond 1e12 s (the caller then
 * formats in Python), PXG_E_INVALID for an index outside a name table. */
typedef struct { const uint32_t* data; int64_t width; } pxg_text_column;
typedef struct {
    int64_t n;
    const int64_t* string_row;
    pxg_text_column text[5];
    const int64_t* start_time;  const double* sampling_rate;  const int64_t* duration;
    const uint8_t* has_summary; const int64_t* num_events;    const int64_t* sequence_length;
    const double* mean_qscore;
    const int32_t* status;      const char* const* status_names;  int32_t n_status;
    const int32_t* label;       const char* const* label_names;   int32_t n_labels;
    const int32_t* barcode;     const int32_t* barcode_score;
    const char* const* barcode_names;  int32_t n_barcode_names;
    const uint8_t* has_polya;   const double* polya_dwell;
} pxg_summary_columns;
int64_t pxg_summary_rows(const pxg_summary_columns* cols, char* out, int64_t cap);

/* ---- compressed samples across PCIe --------------------------------------
 * A read bundle may carry its samples as zig-zag deltas in independent chunks of PXG_Z_CHUNK
 * samples that never span reads; sample 0 of a chunk is in its record.  Two encodings of the
 * deltas, named by the record's `codec`:
 *   PXG_Z_BYTES   one or two bytes each (the variable-byte stage of ONT's VBZ): 128 control bytes
 *                 (bit i: sample i took two bytes) + the data bytes of samples 1 .. len-1;
 *   PXG_Z_PACKED  bit-packed: 128 header bytes = 256 nibbles, the bit width of each group of FOUR
 *                 deltas (code 15 = 16 bits; the chunk's delta 0 is 0; groups past `len` have
 *                 width 0), then the groups' 4 x w bits back to back, least significant bit
 *                 first, padded to a byte.  ~0.96 bytes per sample where PXG_Z_BYTES needs 1.19.
 * pxg_batch_stage_z copies the bytes and the chunk records of a batch to the device and decodes
 * them there into the spare input slot (a wave per chunk), so the link carries ~1 byte per
 * sample instead of 2; the resident batch is the same int16 arena either way.  data_off / dst
 * of the records may be relative to any base: the call takes the bases of the slice it is given.
 * pxg_z_encode(_as) / pxg_z_decode(_n) / pxg_z_count_chunks: host only (libpxghost.so). */
#define PXG_Z_CHUNK 1024
#define PXG_Z_CTRL_BYTES (PXG_Z_CHUNK / 8)
#define PXG_Z_BYTES 0
#define PXG_Z_PACKED 1
struct pxg_z_chunk {
    int64_t data_off;      /* byte offset of the chunk (its control / header bytes) in the encoded stream */
    int64_t dst;           /* sample index of the chunk's first sample in the decoded arena */
    int16_t first;         /* sample 0 */
    int16_t len;           /* samples in the chunk, 1 .. PXG_Z_CHUNK */
    int32_t codec;         /* PXG_Z_BYTES (0: every bundle written before ABI 5) or PXG_Z_PACKED */
};
int64_t pxg_z_count_chunks(int64_t n_reads, const int64_t* offsets);
/* -> bytes written, PXG_E_NOMEM if cap is short (sufficient for either codec:
 * (PXG_Z_CTRL_BYTES + 2 * PXG_Z_CHUNK + 8) * chunks); pxg_z_encode = PXG_Z_BYTES */
int64_t pxg_z_encode(int64_t n_reads, const int16_t* arena, const int64_t* offsets, uint8_t* out,
                     int64_t cap, pxg_z_chunk* chunks);
int64_t pxg_z_encode_as(int64_t n_reads, const int16_t* arena, const int64_t* offsets, uint8_t* out,
                        int64_t cap, pxg_z_chunk* chunks, int32_t codec);
/* pxg_z_decode_n is told the stream's size (reads of a packed chunk are clamped to it);
 * pxg_z_decode: the caller vouches for 128 + 2 048 readable bytes behind every chunk start */
int pxg_z_decode(int64_t n_chunks, const uint8_t* z, const pxg_z_chunk* chunks, int64_t data_base,
                 int64_t dst_base, int16_t* out);
int pxg_z_decode_n(int64_t n_chunks, const uint8_t* z, int64_t z_bytes_or_0, const pxg_z_chunk* chunks,
                   int64_t data_base, int64_t dst_base, int16_t* out);
/* Chunk records come from files: PXG_OK iff the records tile [dst_base, dst_base + n_samples) in
 * order with 1 .. PXG_Z_CHUNK samples each and every chunk lies inside the z_bytes of the stream
 * (PXG_E_INVALID otherwise).  pxg_batch_stage_z runs the same check and refuses the batch; call
 * this before pxg_z_decode on anything read from disk (host only, libpxghost.so). */
int pxg_z_validate(int64_t n_chunks, const pxg_z_chunk* chunks, int64_t data_base, int64_t z_bytes,
                   int64_t dst_base, int64_t n_samples);
/* the resident batch's int16 samples back on the host (n_samples of the last upload / swap):
 * what pxg_batch_stage_z decoded, for a caller that wants the samples themselves (and the tests) */
int pxg_batch_download_samples(pxg_ctx* ctx, int16_t* out);
int pxg_batch_stage_z(pxg_ctx* ctx, int64_t n_reads, const uint8_t* z, int64_t z_bytes,
                      const pxg_z_chunk* chunks, int64_t n_chunks, int64_t data_base, int64_t dst_base,
                      const int64_t* raw_offsets, const pxg_calib* calib,
                      const float* scale_shift_or_null);
/* The same two calls for a batch whose stages read only the first `prefix_limit` samples of a read (0 = all):
 * without the poly(A) stage and the chimera scan nothing reads behind max(scaler_length,
 * segmentation_scan_limit) (signal_analyzer.py:347-349 scans the first segmentation_scan_limit samples; the
 * scaler's head and the barcode window lie inside them), so the rest of a long read stays on the host.  The
 * resident arena keeps the layout of the whole reads (offsets, n_pooled and every result are those of the full
 * call); stages and hooks that read whole reads (PXG_STAGE_POLYA, pxg_batch_unsplit_scan, pxg_batch_event_table,
 * pxg_batch_download_samples) refuse such a batch with PXG_E_STATE when one of its reads is longer than the
 * limit.  pxg_process_batch(_ex) chooses the limit itself from its stage mask and extras. */
int pxg_batch_stage_prefix(pxg_ctx* ctx, int64_t n_reads, const int16_t* raw_arena, const int64_t* raw_offsets,
                           const pxg_calib* calib, const float* scale_shift_or_null, int64_t prefix_limit);
int pxg_batch_stage_z_prefix(pxg_ctx* ctx, int64_t n_reads, const uint8_t* z, int64_t z_bytes,
                             const pxg_z_chunk* chunks, int64_t n_chunks, int64_t data_base, int64_t dst_base,
                             const int64_t* raw_offsets, const pxg_calib* calib,
                             const float* scale_shift_or_null, int64_t prefix_limit);

/* ---- FAST5 input without an HDF5 library (host only, libpxghost.so) ---------
 * What the per-read processor needs from a FAST5 file (fast5_file.py:37-58 get_read_ids,
 * :97-131 Fast5Reader metadata + int16 `Signal', :133-181 basecall summary), read straight from
 * a memory map of the file: single- and multi-read layouts, contiguous / chunked datasets,
 * gzip, shuffle and ONT's VBZ filter (csrc/pxg_h5.cpp lists the supported subset of the HDF5
 * format; anything else is PXG_E_UNSUPPORTED with a message, never a guess).  The signals of a
 * whole batch are decoded on host threads directly into the staging arena pxg_batch_stage
 * copies from.  Per-read problems are DATA (pxg_h5_read_info.status, pxg_h5_load_signals'
 * status array), as in the reference where a bad read is that read's outcome. */
typedef struct pxg_h5 pxg_h5;
typedef struct {
    int32_t status;                 /* 0, or the pxg_error of THIS read (text in `error`)      */
    int32_t bc_present;             /* a Basecall_1D_* group exists                            */
    char read_id[64];
    char channel_number[16];
    char run_id[64];
    char sample_id[128];
    char error[160];
    int64_t duration, start_time;   /* Raw attributes (fast5_file.py:97-109)                   */
    int64_t n_samples;              /* length of the Signal dataset                            */
    pxg_calib calib;                /* channel_id attributes (:110-115)                        */
    int32_t bc_table;               /* 0 none, 1 Move, 2 Guppy Events, 3 albacore, 4 unsupported */
    int32_t bc_block_stride;
    int64_t bc_sequence_length, bc_num_events, bc_first_sample;
    int64_t bc_n_moves, bc_move_sum;/* rows / sum of the move column (-1: no table)            */
    int64_t bc_seq_len;             /* characters of the Fastq sequence line                   */
    double bc_mean_qscore;
} pxg_h5_read_info;
int pxg_h5_open(const char* path, pxg_h5** out);
int pxg_h5_open_mt(const char* path, int32_t threads, pxg_h5** out);   /* read groups walked on host threads */
void pxg_h5_close(pxg_h5* file);
const char* pxg_h5_last_error(void);                       /* of the calling thread */
int64_t pxg_h5_n_reads(const pxg_h5* file);
int pxg_h5_is_multi(const pxg_h5* file);
int pxg_h5_read_id(const pxg_h5* file, int64_t i, char* out, int64_t cap);
/* all ids, '\\n'-separated; returns the bytes needed (nothing written when that exceeds cap) */
int64_t pxg_h5_read_ids(const pxg_h5* file, char* out, int64_t cap);
int pxg_h5_info(const pxg_h5* file, int64_t first, int64_t n, pxg_h5_read_info* out);
int pxg_h5_info_mt(const pxg_h5* file, int64_t first, int64_t n, pxg_h5_read_info* out, int32_t threads);
/* Many files at once, on `threads` host threads (a directory of single-read FAST5: fast5_file.py:60-96 once per read):
 * files[k] / rc[k] as pxg_h5_open leaves them (NULL + the error code; its text in error[160 k .. 160 k + 159] when
 * `error` is given), n_reads[k], multi[k], and -- for a file that holds exactly one read in the single-read layout --
 * first_info[k] as pxg_h5_info(files[k], 0, 1, ..) fills it.  pxg_h5_close_many closes what is not NULL. */
int pxg_h5_open_many(int64_t n, const char* const* paths, int32_t threads, pxg_h5** files, int32_t* rc,
                     int64_t* n_reads, int32_t* multi, pxg_h5_read_info* first_info, char* error_or_null);
void pxg_h5_close_many(int64_t n, pxg_h5* const* files);
/* text = sequence '\n' quality string; move = the Move table / the Events table's move column */
int pxg_h5_basecall(const pxg_h5* file, int64_t i, int64_t text_cap, char* text, int64_t move_cap,
                    uint8_t* move, double* p_model_state_or_null, int32_t* has_p_model_state);
/* The columns of a read's BaseCalled_template/Events table that the per-read processor consumes when the
 * table brings its own events (albacore's 14 columns, fast5_file.py:166-181): start, length, mean, stdv,
 * move, p_model_state as float64 and model_state as text of model_state_cap bytes per row (any may be NULL).
 * info[3 * k + {0, 1, 2}] = class (0 integer, 1 float, 3 text, -1 absent), byte width, signedness of column k
 * in that order (model_state = 6): enough to rebuild the file's dtypes.  Returns the number of rows (0: no
 * Events table); when that exceeds cap_rows nothing is written -- call again with room. */
int64_t pxg_h5_events(const pxg_h5* file, int64_t i, int64_t cap_rows, int32_t* info, double* start,
                      double* length, double* mean, double* stdv, double* move, double* p_model_state,
                      char* model_state, int32_t model_state_cap);
/* int16 samples of many reads (any mix of open files) decoded on `threads` host threads into the
 * caller's (staging) arena: read k -> arena[dst_start[k] .. dst_start[k] + n_samples[k]);
 * status[k] = 0 or that read's own pxg_error */
int pxg_h5_load_signals(int64_t n, const pxg_h5* const* files, const int64_t* index,
                        const int64_t* dst_start, const int64_t* n_samples, int16_t* arena,
                        int32_t threads, int32_t* status);
/* sequences, quality strings and move tables of many reads into columnar arenas (lengths from
 * pxg_h5_info: bc_seq_len, bc_n_moves) */
int pxg_h5_basecall_many(int64_t n, const pxg_h5* const* files, const int64_t* index,
                         const int64_t* seq_start, const int64_t* seq_len, uint8_t* seq_arena,
                         uint8_t* qual_arena, const int64_t* move_start, const int64_t* n_moves,
                         uint8_t* move_arena, int32_t threads, int32_t* status);

#ifdef __cplusplus
}
#endif
#endif /* PXG_H */
