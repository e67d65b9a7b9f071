// pxg_common.h -- internal declarations shared by the HIP translation units of
// libpxg.so (gfx950 only).  Public ABI: include/pxg.h.
//
// Compile flags that are part of the numerics:
//   -ffp-contract=off   every fused multiply-add is an explicit __builtin_fmaf;
//                       a*b+c written as two operations stays two roundings,
//                       exactly as the reference's NumPy/TF expressions do.
//   default f32 denormal mode (preserve) and correctly rounded f32 division.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <condition_variable>
#include <string>
#include <vector>
#include "../../include/pxg.h"

#define PXG_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Canonical float32 activation kit (DESIGN.md "Canonical LSTM arithmetic"),
// bit-exact with oracle/pxo_core.c:
//   sigmoid: cubic Hermite spline, 1024 segments of width 1/16 on [-32, 32),
//            float4 coefficients in LDS (`tab`), u = 16 z, seg = floor(u),
//            s = u - seg (exact), three fma.   9 VALU + one ds_read_b128.
//   tanh(x) = fl(2*sigmoid(2x) - 1).
//   expf (softmax only): Cody-Waite + degree-7 Taylor in fma form.
// ---------------------------------------------------------------------------
#define PXG_SIG_NSEG 1024
#define PXG_SIG_HALF 512

// Segment and position of a look-up argument u (table units, already clamped to the table): floor(u) as the
// index, u - floor(u) as the position -- ONE instruction each.  v_fract_f32 is min(RN(u - floor(u)), 0x1.fffffep-1f)
// and v_cvt_flr_i32_f32 is (int)floor(u) for EVERY float with |u| <= 1024 (tools/ubench/fract_check.hip, all 2.3e9 of
// them, profiles/r04/ubench_fract_check.txt); the clamp below 1 (u a tiny negative number: u + 1 rounds to 1) is part
// of the arithmetic's definition (oracle/pxo_core.c sig_position).  v_floor + v_sub + v_cvt were three.
__device__ __forceinline__ int pxg_sig_segment(float u)
{
    int i;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(u));
    return i;
}
__device__ __forceinline__ float pxg_sig_position(float u)
{
    return __builtin_amdgcn_fractf(u);
}

__device__ __forceinline__ float pxg_sig_lookup(const float4* tab, float z, float zscale,
                                                float zlo, float zhi)
{
    z = __builtin_amdgcn_fmed3f(z, zlo, zhi);
    const float u = z * zscale;
    const float s = pxg_sig_position(u);
    const float4 c = tab[pxg_sig_segment(u) + PXG_SIG_HALF];
    float p = __builtin_fmaf(c.w, s, c.z);
    p = __builtin_fmaf(p, s, c.y);
    return __builtin_fmaf(p, s, c.x);
}

// Same lookup for an argument that already carries the table scale (u = 16 z for
// a sigmoid, u = 32 x for tanh's inner sigmoid): the LSTM kernels fold the power
// of two into the gate weights at upload (exact), see k_lstm.hip cell_update.
__device__ __forceinline__ float pxg_sig_lookup_u(const float4* tab, float u)
{
    u = __builtin_amdgcn_fmed3f(u, -512.0f, 511.99997f);
    const float s = pxg_sig_position(u);
    const float4 c = tab[pxg_sig_segment(u) + PXG_SIG_HALF];
    float p = __builtin_fmaf(c.w, s, c.z);
    p = __builtin_fmaf(p, s, c.y);
    return __builtin_fmaf(p, s, c.x);
}

__device__ __forceinline__ float pxg_sigmoid(const float4* tab, float x)
{
    return pxg_sig_lookup(tab, x, 16.0f, -32.0f, 31.999998f);
}

__device__ __forceinline__ float pxg_tanh(const float4* tab, float x)
{
    const float s = pxg_sig_lookup(tab, x, 32.0f, -16.0f, 15.999999f);
    return __builtin_fmaf(2.0f, s, -1.0f);
}

__device__ __forceinline__ float pxg_expf(float x)
{
    x = __builtin_amdgcn_fmed3f(x, -87.0f, 87.0f);
    const float magic = 12582912.0f;  // 1.5 * 2^23
    float t = __builtin_fmaf(x, 1.44269504088896341f, magic);
    float n = t - magic;
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.98412698412698413e-4f;
    p = __builtin_fmaf(p, r, 1.38888888888888894e-3f);
    p = __builtin_fmaf(p, r, 8.33333333333333322e-3f);
    p = __builtin_fmaf(p, r, 4.16666666666666644e-2f);
    p = __builtin_fmaf(p, r, 1.66666666666666657e-1f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return __int_as_float(__float_as_int(p) + ((int)n << 23));
}

// ---------------------------------------------------------------------------
// log(exp(d) + 1) for d <= 0 in float64: the two-component log-sum-exp of the HMM
// emissions (pomegranate: a + log(exp(b - a) + 1)).  The generic ocml exp + log pair was
// 0.30 of K3's 1.55 ms; this one does the same three steps -- t = exp(d), u = t + 1
// (one rounding, as in the formula), log(u) -- with table-driven kernels for exactly the
// ranges that occur (d <= 0, u in [1, 2]): ~30 float64 operations instead of ~90, each
// step within 1 ulp of libm (tests: emissions vs the oracle's libm values, paths identical).
//   exp:  k = rint(d * 64/ln2), r = d - k ln2/64 (two-part constant), j = k mod 64,
//         t = 2^(k div 64) * T[j] * (1 + expm1_poly5(r)),  T[j] = 2^(j/64)
//   log:  j = rint((u - 1) * 128), c = 1 + j/128, r = (u - c) * (1/c),
//         log u = log c + r + r^2 * poly4(r)
// Tables (host libm, uploaded once): T[64], log c[129], 1/c[129].
// ---------------------------------------------------------------------------
#define PXG_LSE_TAB_DOUBLES (64 + 129 + 129)

__device__ __forceinline__ double pxg_log1pexp(const double* __restrict__ tab, double d)
{
    // exp(d): below -40 the sum t + 1 rounds to 1 and the result is exactly 0, as with libm
    const double dd = d >= -40.0 ? d : -40.0;          // also catches NaN (both operands -inf)
    const double kf = __builtin_rint(dd * 92.332482616893656877);                 // 64 / ln 2
    double r = __builtin_fma(-kf, 0.010830424696244734, dd);       // ln2/64: high part (41 bits, kf * hi exact)
    r = __builtin_fma(-kf, 4.411764150473684e-15, r);              // low part
    const int k = (int)kf;
    const double tj = tab[k & 63];
    double q = __builtin_fma(r, 8.33333333333333322e-03, 4.16666666666666644e-02);
    q = __builtin_fma(r, q, 1.66666666666666657e-01);
    q = __builtin_fma(r, q, 0.5);
    const double e = __builtin_fma(r * r, q, r);                                   // expm1(r)
    double t = __builtin_ldexp(__builtin_fma(tj, e, tj), k >> 6);
    t = d >= -40.0 ? t : 0.0;
    const double u = t + 1.0;
    // log(u), u in [1, 2]
    const double jf = __builtin_rint((u - 1.0) * 128.0);
    const int j = (int)jf;
    const double c = __builtin_fma(jf, 0.0078125, 1.0);
    const double rr = (u - c) * tab[64 + 129 + j];
    double p = __builtin_fma(rr, -1.66666666666666657e-01, 0.2);
    p = __builtin_fma(rr, p, -0.25);
    p = __builtin_fma(rr, p, 3.33333333333333315e-01);
    p = __builtin_fma(rr, p, -0.5);
    return tab[64 + j] + __builtin_fma(rr * rr, p, rr);
}

// Lane select with the mask in an SGPR pair (v_cndmask_b32_e64).  Measured on
// gfx950: a second v_cndmask that re-reads the SAME vcc costs ~8 ns instead of
// ~2 ns, so every multi-word select (f64, packed vectors) goes through these
// helpers with a ballot mask instead of the compiler's vcc form.
__device__ __forceinline__ unsigned pxg_sel_u32(unsigned long long take, unsigned if0, unsigned if1)
{
    unsigned d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(if0), "v"(if1), "s"(take));
    return d;
}
__device__ __forceinline__ float pxg_sel_f32(unsigned long long take, float if0, float if1)
{
    return __uint_as_float(pxg_sel_u32(take, __float_as_uint(if0), __float_as_uint(if1)));
}
// max of two doubles, one v_max_f64 (no NaN can reach the callers: log-probabilities, -inf included)
__device__ __forceinline__ double pxg_max_f64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double pxg_sel_f64(unsigned long long take, double if0, double if1)
{
    const unsigned lo = pxg_sel_u32(take, (unsigned)__double2loint(if0), (unsigned)__double2loint(if1));
    const unsigned hi = pxg_sel_u32(take, (unsigned)__double2hiint(if0), (unsigned)__double2hiint(if1));
    return __hiloint2double((int)hi, (int)lo);
}

// (mask & a) | (~mask & b) in one instruction
__device__ __forceinline__ unsigned pxg_bfi(unsigned mask, unsigned a, unsigned b)
{
    unsigned d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
    return d;
}


// DAQ counts -> pA (fast5_file.py:130-131): float64 product, one cast.
__device__ __forceinline__ float pxg_raw2pa(int16_t raw, double k, double offset)
{
    return (float)(k * ((double)raw + offset));
}

// NumPy float32 pairwise add.reduce of exactly `stride` pA values starting at
// raw[0] (n < 8: sequential; 8..128: eight partial sums + tail), then the
// float32 true-divide of numpy's mean (signal_loader.py:224-225).
// the shipped stride, samples already in registers
__device__ __forceinline__ float pxg_block_mean15(const int16_t (&x)[15], double k, double offset)
{
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = pxg_raw2pa(x[j], k, offset);
    float s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int i = 8; i < 15; i++) s += pxg_raw2pa(x[i], k, offset);
    s = 0.0f + s;
    return s / 15.0f;
}

__device__ __forceinline__ float pxg_block_mean(const int16_t* __restrict__ raw, int stride,
                                                double k, double offset)
{
    float s;
    if (stride == 15) {          // all 15 loads in flight at once, no loop-carried address
        int16_t x[15];
#pragma unroll
        for (int j = 0; j < 15; j++) x[j] = raw[j];
        return pxg_block_mean15(x, k, offset);
    }
    if (stride < 8) {
        s = 0.0f;
        for (int i = 0; i < stride; i++) s += pxg_raw2pa(raw[i], k, offset);
    } else {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = pxg_raw2pa(raw[j], k, offset);
        int i = 8;
        for (; i < stride - (stride % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] += pxg_raw2pa(raw[i + j], k, offset);
        }
        s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < stride; i++) s += pxg_raw2pa(raw[i], k, offset);
    }
    s = 0.0f + s;
    return s / (float)stride;
}

// ---------------------------------------------------------------------------
// host-side context
// ---------------------------------------------------------------------------
struct PxgHmmDev {           // device image of one HMM (by-value kernel argument)
    int n_states;
    int adapter_state, polya_state;
    int left_to_right;       // every edge i->j has j >= i (config order)
    int max_in;              // largest in-degree
    unsigned shift_mask;     // bit k: some edge i->j has j - i == k (left-to-right models)
    int n_mix[PXG_MAX_STATES];
    int order[PXG_MAX_STATES];                 // states in name-sorted order
    int in_src[PXG_MAX_STATES][PXG_MAX_STATES]; // in-edge sources, name-sorted; -1 unused
    double in_logp[PXG_MAX_STATES][PXG_MAX_STATES];
    double log_start[PXG_MAX_STATES];
    double mu[PXG_MAX_STATES][PXG_MAX_MIXTURE];
    double lssp[PXG_MAX_STATES][PXG_MAX_MIXTURE];   // -log(sigma*sqrt(2pi))
    double tss[PXG_MAX_STATES][PXG_MAX_MIXTURE];    // 1/(2 sigma^2)
    double logw[PXG_MAX_STATES][PXG_MAX_MIXTURE];   // log(w/sum w)
};

struct PxgLstmDev {          // one LSTM layer's weights in HBM (Keras layout)
    int input_dim = 0, units = 0;
    float* kernel = nullptr;
    float* recurrent = nullptr;
    float* bias = nullptr;
};

struct PxgDenseDev {
    int in_dim = 0, out_dim = 0;
    float* kernel = nullptr;
    float* bias = nullptr;
};

template <typename T>
struct DevBuf {              // grow-only device arena
    T* p = nullptr;
    size_t cap = 0;
};

struct pxg_ctx {
    pxg_config cfg;
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    std::string err;

    PxgHmmDev hmm[2];
    PxgLstmDev scaler1, scaler2, demux_fwd, demux_bwd, demux_top;
    PxgDenseDev scaler_dense, demux_dense;
    struct {                     // PXG_LSTM_Q8: digit planes of the integer weight matrices in MFMA
        int8_t* scaler_frag = nullptr;   // A-fragment order (k_lstm_q8.hip), 16 bytes per thread and fragment
        int8_t* bidir_frag = nullptr;
        int8_t* top_frag = nullptr;
        int8_t* scaler_frag_lat = nullptr;   // K2's latency form (k_lstm_q8_lat.hip): 8 fragments per gate tile
        int8_t* bidir_frag_lat = nullptr;    // K5a's: [direction][gate tile][3]
        int8_t* top_frag_lat = nullptr;      // K5b's: [gate tile][8]
        float s_scaler1[4] = {}, s_scaler2[4] = {}, s_fwd[4] = {}, s_bwd[4] = {}, s_top[4] = {};   // g * 2^(-p-14) per layer and gate block
        int forced_block_steps = 0;      // host copy of K2's steps-per-task override (trajectory recording, tuning knob)
    } q8;
    double* d_calibration = nullptr;
    float* d_sigtab = nullptr;   // PXG_SIG_NSEG x 4 spline coefficients
    double* d_lsetab = nullptr;  // PXG_LSE_TAB_DOUBLES: tables of pxg_log1pexp

    // resident batch
    int64_t n_reads = 0;
    int64_t n_samples = 0;
    bool have_inject = false;
    DevBuf<int16_t> raw;
    DevBuf<int64_t> offsets;
    DevBuf<pxg_calib> calib;
    DevBuf<float> inject;        // n x 2
    // spare input slot (pxg_batch_stage / pxg_batch_swap) and its copy stream
    struct {
        DevBuf<int16_t> raw;
        DevBuf<int64_t> offsets;
        DevBuf<pxg_calib> calib;
        DevBuf<float> inject;
        int64_t n_reads = 0, n_samples = 0;
        bool have_inject = false, staged = false;
        double rate_min = 0.0, rate_max = 0.0;
        int64_t limit = 0;           // samples of each read that were copied (0: all)
        DevBuf<uint8_t> z;           // encoded samples of a pxg_batch_stage_z batch ...
        DevBuf<pxg_z_chunk> zchunks; // ... and their chunk records, decoded into `raw` on the copy stream
    } spare;
    hipStream_t copy_stream = nullptr;
    // The window scan of the pseudo-fusion filter (K7a / K7b) reads only what the segmentation stage left behind, and
    // so does poly(A) (K6): a scan call that follows a run with the poly(A) stage puts its kernels on this stream,
    // with two gates: the block means (K7a, VALU-bound) may start once the segmentation is done, beside the barcode
    // kernels; the window scan (K7b, latency-bound) behind the event recorded just before K6's launch -- two
    // latency-bound kernels side by side on the chip instead of one after the other, and K7a out of their way (started
    // at the same gate it took 2.75 ms instead of 0.53).  PXG_NO_SCAN_OVERLAP=1 at pxg_create: everything on `stream`.
    hipStream_t scan_stream = nullptr;
    hipEvent_t ev_segmented = nullptr, ev_scan_gate = nullptr, ev_scan_done = nullptr;
    bool scan_gate_set = false, scan_overlap = true;
    // round 6: K6 (poly(A)) needs nothing the barcode stage writes -- with both stages in a run it starts behind K3 on a
    // stream of its own, beside K4 / K5a / K5b, and the main stream waits for it before the records are built
    // (PXG_NO_POLYA_OVERLAP=1: behind K5b on the main stream, as until round 5)
    hipStream_t polya_stream = nullptr;
    hipEvent_t ev_polya_done = nullptr;
    bool polya_overlap = true;
    hipEvent_t ev_staged = nullptr;
    hipEvent_t ev_run_done[2] = { nullptr, nullptr };   // last run on the resident / the spare inputs
    bool run_recorded[2] = { false, false };
    int cur = 0;                 // which of the two belongs to the resident inputs
    DevBuf<float> head;          // n x head_width
    DevBuf<float> pred;          // n x 2
    DevBuf<float> ss;            // n x 2
    DevBuf<int32_t> status;      // n
    DevBuf<int32_t> segs;        // n x 2 x PXG_N_SEGMENTS
    DevBuf<unsigned long long> unsplit_q;   // K7b unit queue
    DevBuf<char> vit_bp;         // K3 back-pointer fields, [block][chunk][64 lanes]
    DevBuf<int32_t> idx_scaler;  // compacted read indices
    DevBuf<int32_t> order, order_hist;   // reads by length class (longest first) of the resident batch; class cursors
    bool length_order = true;    // PXG_NO_LENGTH_ORDER=1 at pxg_create: blocks take the reads in batch order
    DevBuf<unsigned> scaler_traj;   // PXG_LSTM_Q8: tile states of the zero-input scaler network every 4 steps
    bool prefix_skip = true;     // PXG_NO_PREFIX_SKIP=1 at pxg_create: K2 runs every read from step 0
    DevBuf<int32_t> idx_demux;
    DevBuf<int32_t> counters;    // [0] scaler count, [1] demux count
    DevBuf<float> win;           // n x trim
    DevBuf<float> bidir;         // n x trim x (Hf+Hb), permuted layout
    DevBuf<float> probs;         // n x PXG_MAX_CLASSES
    DevBuf<int> lstm_q;          // time-sliced LSTM: [0] task counter, [2+] per-tile progress
    DevBuf<int> lstm_err;        // time-sliced LSTM: sticky error flag
    DevBuf<float> lstm_state;    // time-sliced LSTM: tile states in flight between step blocks
    DevBuf<int> demux_q;         // time-sliced demux kernels: two queues (bidir, top)
    DevBuf<float> demux_state;   // time-sliced demux kernels: two alternating state slots per tile
    bool timeslice_used = false;
    DevBuf<pxg_read_result> results;
    DevBuf<char> polya_ev;       // event scratch, [wave][event][lane]
    DevBuf<int32_t> polya_over;  // [0] reads to re-run, [1] most event rows asked for, [2] spike rows handed out, [4..] their ids
    DevBuf<int32_t> polya_retry; // the ids a retry launch works through
    DevBuf<int32_t> polya_order; // first pass: reads by poly(A) length class, longest first (+ 1 024 class cursors behind them)
    bool polya_unsettled = false;   // K6 ran and its overflow list has not been looked at yet
    uint32_t last_stage_mask = 0;
    DevBuf<int32_t> polya_out;   // n x 8: called, n_spikes, dwell, begin lo/hi, end lo/hi, first spike row
    DevBuf<pxg_polya_spike> spikes;   // spike arena of the batch: rows handed out by K6 with one atomic per read
    int64_t spike_rows = 0;      // rows handed out by the settled run
    int64_t resident_limit = 0;  // the resident batch holds only this many samples of each read (0: all)
    int64_t longest_read = 0, spare_longest_read = 0;   // samples of the longest read (host copy)
    std::mutex mt_err;
    // Small pxg_process_batch calls that arrive while the pipeline is full are run as ONE batch (pxg_api.hip,
    // merged_process): the reference hands its workers 128 reads at a time (commandline.py:402), and a 128-read
    // batch costs the latency floor of the kernels (5.5 ms) like a 4 000-read one.
    struct MergeItem {
        int64_t n; const int16_t* arena; const int64_t* off; const pxg_calib* cal; pxg_read_result* out;
        // arena == NULL: the samples arrive encoded (pxg_batch_stage_z's arguments)
        const uint8_t* z; int64_t z_bytes; const pxg_z_chunk* chunks; int64_t n_chunks, data_base, dst_base;
        int rc; bool done;
    };
    struct {
        std::mutex m;
        std::condition_variable cv;
        std::vector<MergeItem*> pending;
        int64_t pending_reads = 0;           // reads of `pending` (a group is capped at PXG_MERGE_MAX_GROUP_READS)
        uint32_t mask = 0;
        bool leader = false;                 // a caller is waiting for the spare slot on behalf of `pending`
        std::vector<int64_t> h_off;          // merged offsets / calibration / records of the group in flight
        std::vector<pxg_calib> h_cal;
        std::vector<pxg_read_result> h_out;
        int64_t groups = 0, calls = 0;       // statistics: groups run, calls they carried
    } merge;
    // Device -> host copies whose destination is ordinary (pageable) memory go through this page-locked buffer and a
    // memcpy: the HIP runtime would otherwise pin the destination's pages on the fly for anything above ~1 MB and keep
    // that mapping cached -- with destinations that are allocated and freed around every call (NumPy result arrays,
    // std::vectors) a cached mapping can outlive its pages (heap trimmed, address reused), and the next copy to that
    // address faults on the GPU (seen once in ~40 runs of bench.py --workload full: "Memory access fault by GPU" at a
    // heap address).  Destinations the caller page-locked (pxg_host_register) are written directly.
    void* h_bounce = nullptr;
    size_t h_bounce_bytes = 0;
    std::mutex mt_bounce;        // the stage hooks take no other lock: two threads may download at once
    // page-locked chunks every big PAGEABLE host -> device copy goes through (pxg_h2d_big: a pair per copying thread)
    void* h_up[16] = {};
    hipEvent_t ev_up[16] = {};
    bool up_busy[16] = {};
    std::mutex mt_up;
    // ... and the small host arrays of a batch (offsets, calibration, injected scaling, chunk-record-free metadata)
    // are copied into page-locked mirrors first and sent from there: [0] the staging calls (one stage at a time:
    // mt_stage / the caller's own order), [1] pxg_batch_upload.
    struct HostMirror { void* p = nullptr; size_t cap = 0; } h_meta[3][4];     // [spare slot | resident batch | scan hooks][array]
    bool merge_small_calls = true;           // PXG_NO_CALL_MERGE=1 at pxg_create: every call is its own batch
    std::mutex mt_stage, mt_run; // pxg_process_batch(_ex) from several host threads: spare slot / resident batch
    bool polya_ran = false;
    DevBuf<int64_t> ev_first, ev_off;   // K7: per-read first sample / event offsets
    DevBuf<int64_t> ev_tstart;          // K7: `start' column of tables that bring their own events (albacore)
    DevBuf<float> ev_mean, ev_scaled;   // K7: Guppy block means
    DevBuf<char> unsplit_scr;           // K7: back-pointer + path scratch
    DevBuf<char> unsplit_cand;          // K7: per-window candidates
    DevBuf<int64_t> unit_off;           // K7: first (read, window) unit of each read
    DevBuf<int32_t> n_win;              // K7: windows per read
    DevBuf<int64_t> unsplit_iv;         // compact candidate list (pairs), all reads back to back
    DevBuf<int64_t> unsplit_ivoff;      // n + 1: first pair of each read
    DevBuf<int32_t> unsplit_cnt;
    double rate_min = 0.0, rate_max = 0.0;   // sampling-rate range of the resident batch (host copy)

    hipEvent_t ev_start[PXG_N_TIMERS];
    hipEvent_t ev_stop[PXG_N_TIMERS];
    bool ev_used[PXG_N_TIMERS];
    int64_t launches[PXG_N_TIMERS];
};

// the message of the last failure; written under a lock (pxg_process_batch may run on several
// host threads of one context)
static inline void pxg_set_err(pxg_ctx* ctx, std::string msg)
{
    std::lock_guard<std::mutex> g(ctx->mt_err);
    ctx->err = std::move(msg);
}

#define PXG_HIP(ctx, call)                                                          \
    do {                                                                            \
        hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess) {                                                    \
            pxg_set_err((ctx), std::string(#call) + ": " + hipGetErrorString(e__)); \
            return PXG_E_HIP;                                                       \
        }                                                                           \
    } while (0)

template <typename T>
static inline int pxg_reserve(pxg_ctx* ctx, DevBuf<T>& b, size_t n)
{
    if (n <= b.cap && b.p) return PXG_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc((void**)&b.p, want * sizeof(T));
    if (e != hipSuccess) {
        pxg_set_err(ctx, std::string("hipMalloc: ") + hipGetErrorString(e));
        return PXG_E_NOMEM;
    }
    b.cap = want;
    return PXG_OK;
}

// ---- kernel launchers (defined in the k_*.hip units) -----------------------
int pxg_launch_head_pool(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                         const pxg_calib* cal, float* head, int32_t* status);
int pxg_launch_raw_to_pa(pxg_ctx* ctx, int64_t n, const int16_t* raw, const pxg_calib* cal,
                         float* out);
int pxg_launch_pool_scale(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                          const pxg_calib* cal, const float* ss, const int64_t* poff,
                          float* out);
int pxg_launch_pooled_stretch(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                              const pxg_calib* cal, const float* ss, const int64_t* first,
                              const int64_t* ooff, float* out, int* flag);
int pxg_launch_scaler_transform(pxg_ctx* ctx, int64_t n, const float* pred, float* ss,
                                int32_t* status, const int32_t* idx, const int32_t* count);
// segmentation of raw reads (pool + scale + Viterbi + run summary)
int pxg_launch_segment_raw(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                           const pxg_calib* cal, const float* ss, const float* head_or_null,
                           const int32_t* status, int32_t* segs, const int32_t* order = nullptr);
// Viterbi on already pooled float signals (test hook)
int pxg_launch_viterbi_f32(pxg_ctx* ctx, int which, int64_t n, const float* sig,
                           const int64_t* off, int max_steps, int32_t* segs, double* logp);
int pxg_launch_barcode_window_raw(pxg_ctx* ctx, int64_t n, const int16_t* raw,
                                  const int64_t* off, const pxg_calib* cal, const float* ss,
                                  const int32_t* status, const int32_t* segs, float* win,
                                  int32_t* idx_demux, int32_t* counter);
int pxg_launch_barcode_window_f32(pxg_ctx* ctx, int64_t n, const float* sig,
                                  const int64_t* off, float* win, int8_t* pushed);
int pxg_launch_compact_scaler(pxg_ctx* ctx, int64_t n, const int32_t* status, const int32_t* order, int32_t* idx,
                              int32_t* counter);
// reads by length class, longest first (k_signal.hip)
int pxg_launch_length_order(pxg_ctx* ctx, int64_t n, const int64_t* off, int32_t* order);
// `off` (or null): sample offsets of the reads `head` was pooled from -- lets K2 skip the zero-pad prefix
int pxg_launch_scaler_lstm(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx,
                           const int32_t* count, const float* head, float* pred, const int64_t* off = nullptr);
int pxg_launch_demux_lstm(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx,
                          const int32_t* count, const float* win, float* bidir, float* probs,
                          int timer_a, int timer_b);
int pxg_launch_reset_batch(pxg_ctx* ctx, int64_t n);
// device -> host on the context's stream, synchronised: directly into page-locked destinations, through the context's
// page-locked bounce buffer otherwise (pxg_api.hip; callers hold the run lock or are the only user of the context)
int pxg_d2h_sync(pxg_ctx* ctx, void* dst, const void* src, size_t bytes);
// host -> device of a small array through page-locked mirror h_meta[set][which] (asynchronous on `st`; the mirror is
// reused by the next call of the same set, which by construction starts after this copy has completed)
int pxg_h2d_meta(pxg_ctx* ctx, int set, int which, void* dst, const void* src, size_t bytes, hipStream_t st);
int pxg_launch_z_decode(pxg_ctx* ctx, hipStream_t stream, int64_t n_chunks, const uint8_t* z, int64_t z_bytes, const pxg_z_chunk* chunks,
                        int64_t data_base, int64_t dst_base, int16_t* out, const int64_t* off = nullptr, int64_t n_reads = 0,
                        int64_t prefix_limit = 0);
int pxg_launch_finalize(pxg_ctx* ctx, int64_t n, uint32_t stage_mask);
int pxg_lstm_upload(pxg_ctx* ctx);   // shape checks
// PXG_LSTM_Q8 (k_lstm_q8.hip)
int pxg_q8_upload(pxg_ctx* ctx);
void pxg_q8_free(pxg_ctx* ctx);
int pxg_launch_scaler_lstm_q8(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx, const int32_t* count,
                              const float* head, float* pred, const int64_t* off);
int pxg_launch_scaler_lstm_q8_lat(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx, const int32_t* count,
                                  const float* head, float* pred);      // <= 4 x #CU reads (k_lstm_q8_lat.hip)
int pxg_launch_demux_lstm_q8_lat(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx, const int32_t* count,
                                 const float* win, float* bidir, float* probs, int timer_a, int timer_b);
int pxg_q8_scaler_trajectory(pxg_ctx* ctx);   // zero-input states of the scaler network (prefix skip), once per context
int pxg_launch_demux_lstm_q8(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx, const int32_t* count,
                             const float* win, float* bidir, float* probs, int timer_a, int timer_b);
int pxg_h2d_big(pxg_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st);
int pxg_polya_supported(pxg_ctx* ctx);
int pxg_launch_polya(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                     const pxg_calib* cal, const float* ss, const int32_t* status,
                     const int32_t* segs, int32_t* pout, DevBuf<pxg_polya_spike>& spikes);
int pxg_polya_settle(pxg_ctx* ctx, int64_t n, int64_t longest_read, const int16_t* raw, const int64_t* off,
                     const pxg_calib* cal, const float* ss, const int32_t* status, const int32_t* segs,
                     int32_t* pout, DevBuf<pxg_polya_spike>& spikes, int64_t* retried, int64_t* spike_rows);
int pxg_polya_collect_spikes(pxg_ctx* ctx, int64_t n, const int32_t* pout, const DevBuf<pxg_polya_spike>& spikes,
                             int64_t spike_rows, int64_t cap_rows, pxg_polya_spike* out, int64_t* offsets);
int pxg_launch_detect_events(pxg_ctx* ctx, int64_t n, const float* sig, const int64_t* off,
                             int64_t cap, void* evbuf, int64_t* n_events);

#ifndef PXG_PA_LANES
#define PXG_PA_LANES 4    // reads per wave in k_polya / k_detect_events (16 lanes each)
#endif

// K7: Guppy event means + pseudo-fusion window scan (k_unsplit.hip)
int pxg_launch_guppy_event_means(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                                 const pxg_calib* cal, const float* ss, const int64_t* first,
                                 const int64_t* ev_off, int stride, float* mean, float* scaled,
                                 float* stdv_or_null = nullptr);
int pxg_launch_unsplit_plan(pxg_ctx* ctx, int64_t n, const pxg_calib* cal, const int32_t* status,
                            const int32_t* segs, const int64_t* first_sample, const int64_t* ev_off,
                            int stride, int32_t* n_win, const int64_t* ev_start = nullptr);
int pxg_launch_scale_event_means(pxg_ctx* ctx, int64_t n, int64_t n_events, const int64_t* ev_off, const float* ss,
                                 const float* mean, float* scaled);
int pxg_launch_exclusive_scan(pxg_ctx* ctx, int64_t n, const int32_t* in, int64_t* out /* n + 1 */);
size_t pxg_unsplit_scratch_bytes(const pxg_ctx* ctx, int64_t units_bound, int tmax);
int pxg_unsplit_cand_slots(const pxg_ctx* ctx, int tmax, int stride);
size_t pxg_unsplit_cand_bytes(int64_t units_bound, int wcand);
int pxg_launch_unsplit_scan(pxg_ctx* ctx, int64_t n, int64_t units_bound, int tmax, const pxg_calib* cal,
                            const int32_t* status, const int32_t* segs, const int64_t* first_sample,
                            const int64_t* ev_off, const int64_t* unit_off, const float* scaled, int stride,
                            void* scratch, void* candbuf, int wcand, int32_t* out_cnt, const int64_t* ev_start = nullptr);
int pxg_launch_unsplit_gather(pxg_ctx* ctx, int64_t n, int64_t units_bound, const int64_t* unit_off,
                              const void* candbuf, int wcand, const int32_t* out_cnt, const int64_t* iv_off,
                              int64_t cap, int64_t* out_iv);

void pxg_timer_begin(pxg_ctx* ctx, int t);
void pxg_timer_end(pxg_ctx* ctx, int t);
