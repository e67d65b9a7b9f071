// k_unsplit.hip -- K7: Guppy event table (a18) and the window scan of the
// pseudo-fusion filter (a19).
//   a18  fast5_file.py:210-230  medfilt(5) of pA[first, first+15n) -> block means
//        signal_analyzer.py:318 scaled_mean = fl(fl(scale*mean)+shift)
//   a19  signal_analyzer.py:366-418: 8 s windows stepped 3 s over the event
//        table, Viterbi of the "unsplit" HMM (has back-edges), leader*/adapter run
//        analysis with duration cut-offs -> candidate in-read adapters.
//
// The windows of a read are independent, so the parallel unit is the (read,
// window) pair: a plan kernel counts each read's windows, a one-block scan turns
// the counts into unit offsets (on the device: the host never waits inside the
// call), one 8-lane group scans one window, two gather passes restore the
// reference's append order and compact the candidates of all reads into one CSR
// list.  The unsplit HMM is not left-to-right, so this Viterbi keeps real back
// pointers (see k_unsplit_scan).
#include <type_traits>
#include "pxg_common.h"

#define UN_READS 8
#define UN_CHUNK 16
#define UN_EM_STRIDE (UN_CHUNK * PXG_MAX_STATES + 8)
// Candidate adapters per window: NOT a constant.  The candidates of a window are disjoint
// stretches of it, each at least min(loosen, strict)_full_length long and at least two blocks
// (an adapter run and what separates it from the next), so a window of tmax blocks holds at most
// pxg_unsplit_cand_slots() of them -- the slots are sized from the config and the batch's
// sampling rates, like tmax itself, and a window can never overflow them (14 for the preset).

// ---------------------------------------------------------------------------
// a18: one thread per event block
// ---------------------------------------------------------------------------
__device__ __forceinline__ float med5(float a, float b, float c, float d, float e)
{
    // median of five by a 7-exchange network
    float t;
#define CS(x, y) t = fminf(x, y); y = fmaxf(x, y); x = t;
    CS(a, b) CS(d, e) CS(a, d) CS(b, e) CS(b, c) CS(c, d) CS(b, c)
#undef CS
    return c;
}

// NumPy's pairwise float32 sum of a row of `stride` values (the reduction of mean / std over
// the contiguous axis of a [events, stride] array)
__device__ __forceinline__ float np_row_sum(const float (&blk)[16], int stride)
{
    float s;
    if (stride < 8) {
        s = 0.0f;
        for (int j = 0; j < stride; j++) s += blk[j];
    } else {
        float rsum[8];
#pragma unroll
        for (int j = 0; j < 8; j++) rsum[j] = blk[j];
        int i = 8;
        for (; i < stride - (stride % 8); i += 8)
            for (int j = 0; j < 8; j++) rsum[j] += blk[i + j];
        s = ((rsum[0] + rsum[1]) + (rsum[2] + rsum[3])) + ((rsum[4] + rsum[5]) + (rsum[6] + rsum[7]));
        for (; i < stride; i++) s += blk[i];
    }
    return 0.0f + s;
}

// SD: also the per-event standard deviation of the dumped event table (fast5_file.py:227,
// float32 NumPy _var: fl(x - mean), fl(d * d), the same row sum, / stride, sqrt)
template <bool SD>
__global__ void k_guppy_event_means(int64_t n_reads, const int16_t* __restrict__ raw,
                                    const int64_t* __restrict__ off, const pxg_calib* __restrict__ cal,
                                    const float* __restrict__ ss, const int64_t* __restrict__ first_sample,
                                    const int64_t* __restrict__ ev_off, int stride,
                                    float* __restrict__ mean, float* __restrict__ scaled,
                                    float* __restrict__ stdv)
{
    const int64_t r = blockIdx.x;            // reads on x: gridDim.y stops at 65535
    if (r >= n_reads) return;
    const int64_t n_ev = ev_off[r + 1] - ev_off[r];
    const int64_t n_raw = off[r + 1] - off[r];
    const int64_t first = first_sample[r];
    const int64_t last = first + (int64_t)stride * n_ev;
    const int64_t len = (last > n_raw ? n_raw : last) - first;     // filtered samples available
    const pxg_calib c = cal[r];
    const double k = c.range / c.digitisation;
    const float scale = ss[2 * r], shift = ss[2 * r + 1];
    const int16_t* base = raw + off[r] + first;
    for (int64_t e = blockIdx.y * (int64_t)blockDim.x + threadIdx.x; e < n_ev;
         e += (int64_t)gridDim.y * blockDim.x) {
        float blk[16];
        // pA of the stride + 4 samples this block's medians touch, zero outside [0, len)
        float w[20];
        const int64_t q0 = e * stride;
        if (stride == 15 && q0 >= 2 && q0 + 17 < len) {      // interior block: 19 loads in flight at once
            int16_t x[19];
#pragma unroll
            for (int j = 0; j < 19; j++) x[j] = base[q0 + j - 2];
#pragma unroll
            for (int j = 0; j < 19; j++) w[j] = pxg_raw2pa(x[j], k, c.offset);
            w[19] = 0.0f;
        } else {
#pragma unroll
            for (int j = 0; j < 20; j++) {
                const int64_t p = q0 + j - 2;
                w[j] = (j < stride + 4 && p >= 0 && p < len) ? pxg_raw2pa(base[p], k, c.offset) : 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j < stride)
                blk[j] = (q0 + j >= len) ? __builtin_nanf("") : med5(w[j], w[j + 1], w[j + 2], w[j + 3], w[j + 4]);
        }
        // NumPy pairwise float32 sum of `stride` (= 15) values, then / stride
        const float s = np_row_sum(blk, stride);
        const float m = s / (float)stride;
        mean[ev_off[r] + e] = m;
        const float y = scale * m;
        scaled[ev_off[r] + e] = y + shift;
        if (SD) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (j < stride) {
                    const float d = blk[j] - m;
                    blk[j] = d * d;
                }
            }
            const float v = np_row_sum(blk, stride) / (float)stride;
            stdv[ev_off[r] + e] = sqrtf(v);          // (correctly rounded: the default of hipcc)
        }
    }
}

int pxg_launch_guppy_event_means(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                                 const pxg_calib* cal, const float* ss, const int64_t* first,
                                 const int64_t* ev_off, int stride, float* mean, float* scaled,
                                 float* stdv_or_null)
{
    if (n <= 0) return PXG_OK;
    if (stride < 1 || stride > 16) {
        pxg_set_err(ctx, "block_stride must be 1..16");
        return PXG_E_UNSUPPORTED;
    }
    if (stdv_or_null)
        hipLaunchKernelGGL(k_guppy_event_means<true>, dim3((unsigned)n, 16), dim3(256), 0, ctx->stream, n, raw, off,
                           cal, ss, first, ev_off, stride, mean, scaled, stdv_or_null);
    else
        hipLaunchKernelGGL(k_guppy_event_means<false>, dim3((unsigned)n, 16), dim3(256), 0, ctx->stream, n, raw, off,
                           cal, ss, first, ev_off, stride, mean, scaled, (float*)nullptr);
    return PXG_OK;
}

// scaled_mean of a table that brings its own event means (albacore Events): np.poly1d(float32[scale, shift])
// (mean) = fl(fl(scale * mean) + shift), signal_analyzer.py:318.  One thread per event.
__global__ void k_scale_event_means(int64_t n_reads, const int64_t* __restrict__ ev_off, const float* __restrict__ ss,
                                    const float* __restrict__ mean, float* __restrict__ scaled)
{
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= ev_off[n_reads]) return;
    int64_t lo = 0, hi = n_reads;              // largest r with ev_off[r] <= e
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (ev_off[mid] <= e) lo = mid; else hi = mid; }
    const float y = ss[2 * lo] * mean[e];
    scaled[e] = y + ss[2 * lo + 1];
}

int pxg_launch_scale_event_means(pxg_ctx* ctx, int64_t n, int64_t n_events, const int64_t* ev_off, const float* ss,
                                 const float* mean, float* scaled)
{
    if (n <= 0 || n_events <= 0) return PXG_OK;
    hipLaunchKernelGGL(k_scale_event_means, dim3((unsigned)((n_events + 255) / 256)), dim3(256), 0, ctx->stream, n, ev_off,
                       ss, mean, scaled);
    return PXG_OK;
}

// ---------------------------------------------------------------------------
// a19: window scan
// ---------------------------------------------------------------------------
struct UnsplitParams {
    double window_size, window_step, strict_duration;
    double strict_full, strict_dna, loosen_full, loosen_dna;
    int adapter_state, ll_state, lh_state, seg_adapter_state;
    int stride;            // Guppy block stride
    int pool_stride;       // rough_signal_stride (payload start units)
};

__device__ __forceinline__ double un_emission(const PxgHmmDev& H, const double* lsetab, int s, double x)
{
    double lp;
    {
        const double d = x - H.mu[s][0];
        lp = H.lssp[s][0] - (d * d) * H.tss[s][0];
    }
    if (H.n_mix[s] > 1) {
        lp = lp + H.logw[s][0];
        for (int k = 1; k < H.n_mix[s]; k++) {
            const double d = x - H.mu[s][k];
            const double l = (H.lssp[s][k] - (d * d) * H.tss[s][k]) + H.logw[s][k];
            // pair_lse(a, b) without lane-divergent branches: the items of a wave disagree on
            // a > b, and the if / else form ran exp + log TWICE per item (both sides under exec
            // masks).  m + log(exp(lo - m) + 1) is the same expression on either side; the
            // infinities are selects (one -inf: exp(-inf) = 0, log(1) = 0, m + 0 = m exactly).
            const double a = lp, b = l;
            const bool agb = a > b;
            const double m = agb ? a : b, lo = agb ? b : a;
            const double r = m + pxg_log1pexp(lsetab, lo - m);
            lp = (m == -__builtin_inf()) ? m : ((m == __builtin_inf()) ? m : r);
        }
    }
    return lp;
}

__device__ __forceinline__ double un_shfl_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src);
    hi = __shfl(hi, src);
    return __hiloint2double(hi, lo);
}

// per-read window geometry (signal_analyzer.py:369-388)
struct UnsplitGeom {
    int64_t first, n_ev, payload_start, last_end, window_size, window_step, strict_duration;
    int64_t cut_total[2], cut_adapter[2];
    const int64_t* starts;     // the table's own `start' column (albacore Events, ascending); null: Guppy blocks
    bool valid;
};

// sample at which event k of the frame starts / ends (signal_analyzer.py:321-324: end = next start,
// the last event ends one sample after it starts)
__device__ __forceinline__ int64_t un_ev_start(const UnsplitGeom& g, int stride, int64_t k)
{
    return g.starts ? g.starts[k] : g.first + (int64_t)stride * k;
}
__device__ __forceinline__ int64_t un_ev_end(const UnsplitGeom& g, int stride, int64_t k)
{
    return k == g.n_ev - 1 ? un_ev_start(g, stride, k) + 1 : un_ev_start(g, stride, k + 1);
}

__device__ __forceinline__ UnsplitGeom unsplit_geometry(int64_t r, int64_t n_reads, const UnsplitParams& P,
                                                        const pxg_calib* cal, const int32_t* status,
                                                        const int32_t* segs, const int64_t* first_sample,
                                                        const int64_t* ev_off, const int64_t* ev_start)
{
    UnsplitGeom g;
    g.starts = nullptr;
    g.valid = r < n_reads && status[r] == PXG_ST_OKAY;
    g.first = g.n_ev = g.payload_start = g.last_end = 0;
    g.window_size = 0; g.window_step = 1; g.strict_duration = 0;
    g.cut_total[0] = g.cut_total[1] = g.cut_adapter[0] = g.cut_adapter[1] = 0;
    if (!g.valid) return g;
    const int32_t* sf = segs + r * 2 * PXG_N_SEGMENTS;
    const int a_last = sf[PXG_N_SEGMENTS + P.seg_adapter_state];
    g.n_ev = ev_off[r + 1] - ev_off[r];
    if (sf[P.seg_adapter_state] < 0 || g.n_ev <= 0) { g.valid = false; return g; }
    const double rate = cal[r].sampling_rate;
    g.first = ev_start ? 0 : first_sample[r];
    g.starts = ev_start ? ev_start + ev_off[r] : nullptr;
    g.payload_start = (int64_t)(a_last + 1) * P.pool_stride;          // :369
    g.last_end = un_ev_start(g, P.stride, g.n_ev - 1) + 1;            // events.iloc[-1]['end']
    g.window_size = (int64_t)(P.window_size * rate);                  // :374-383 int(config * rate)
    g.window_step = (int64_t)(P.window_step * rate);
    g.strict_duration = (int64_t)(P.strict_duration * rate);
    g.cut_total[0] = (int64_t)(P.loosen_full * rate); g.cut_total[1] = (int64_t)(P.strict_full * rate);
    g.cut_adapter[0] = (int64_t)(P.loosen_dna * rate); g.cut_adapter[1] = (int64_t)(P.strict_dna * rate);
    if (g.window_step <= 0) g.valid = false;
    return g;
}

// events with left <= start <= left + window_size (both ends inclusive)
__device__ __forceinline__ bool unsplit_window(const UnsplitGeom& g, int stride, int64_t left,
                                               int64_t& k0, int64_t& k1)
{
    if (g.starts) {            // first event with start >= left, last event with start <= left + window
        const int64_t right = left + g.window_size;
        int64_t lo = 0, hi = g.n_ev;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (g.starts[mid] < left) lo = mid + 1; else hi = mid; }
        k0 = lo;
        hi = g.n_ev;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (g.starts[mid] <= right) lo = mid + 1; else hi = mid; }
        k1 = lo - 1;
        return k1 >= k0;
    }
    k0 = (left - g.first) <= 0 ? 0 : (left - g.first + stride - 1) / stride;
    k1 = (left + g.window_size - g.first) < 0 ? -1 : (left + g.window_size - g.first) / stride;
    if (k1 > g.n_ev - 1) k1 = g.n_ev - 1;
    return k1 >= k0;
}

// plan: number of windows the reference visits before its first empty block
__global__ void k_unsplit_plan(int64_t n_reads, UnsplitParams P, const pxg_calib* __restrict__ cal,
                               const int32_t* __restrict__ status, const int32_t* __restrict__ segs,
                               const int64_t* __restrict__ first_sample,
                               const int64_t* __restrict__ ev_off, const int64_t* __restrict__ ev_start,
                               int32_t* __restrict__ n_win)
{
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const UnsplitGeom g = unsplit_geometry(r, n_reads, P, cal, status, segs, first_sample, ev_off, ev_start);
    int cnt = 0;
    if (g.valid) {
        for (int64_t left = g.payload_start; left < g.last_end; left += g.window_step) {
            int64_t k0, k1;
            if (!unsplit_window(g, P.stride, left, k0, k1)) break;               // :387-388
            cnt++;
        }
    }
    n_win[r] = cnt;
}

// out[0] = 0, out[i + 1] = out[i] + max(in[i], 0): one block, each thread a contiguous
// slice, wave + LDS scan of the slice totals.  n is a batch's read count (<= ~1e6), so a
// single block is a few microseconds and needs no second kernel.
__global__ __launch_bounds__(1024) void k_exclusive_scan(int64_t n, const int32_t* __restrict__ in,
                                                         int64_t* __restrict__ out)
{
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
    int64_t s = 0;
    for (int64_t i = lo; i < hi; i++) s += in[i] > 0 ? in[i] : 0;
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {           // Hillis-Steele over the 1024 slice totals
        const int64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int64_t run = part[t] - s;
    for (int64_t i = lo; i < hi; i++) {
        out[i] = run;
        run += in[i] > 0 ? in[i] : 0;
    }
    if (t == 1023) out[n] = part[1023];
}

int pxg_launch_exclusive_scan(pxg_ctx* ctx, int64_t n, const int32_t* in, int64_t* out)
{
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, ctx->stream, n, in, out);
    return PXG_OK;
}

// scan: one 8-lane group per (read, window) unit; unit_off[r] = first unit of read r,
// unit_off[n_reads] = number of units (read on the device: the grid is persistent).
// Back pointers: the 3-bit sources of a group's states are OR-combined over its 8
// lanes (three DPP steps) into one 24-bit table word per step, so the traceback
// of a group is a chain of bit-field extractions -- no cross-lane read on the
// serial path.
// The leader/adapter run analysis runs inside the traceback (backwards: an
// adapter run, then the leader runs that precede it), so no path is stored;
// candidates come out last-first and k_unsplit_gather reverses them.
template <int NIN>
__global__ __launch_bounds__(64) void k_unsplit_scan(
    int64_t n_reads, int tmax, PxgHmmDev H, UnsplitParams P, const pxg_calib* __restrict__ cal,
    const int32_t* __restrict__ status, const int32_t* __restrict__ segs,
    const int64_t* __restrict__ first_sample, const int64_t* __restrict__ ev_off,
    const int64_t* __restrict__ ev_start, const int64_t* __restrict__ unit_off, const float* __restrict__ scaled,
    unsigned* __restrict__ bpbuf /* [wave][tmax][8 groups] */,
    int64_t* __restrict__ cand /* n_units x wcand x 2 */, int32_t* __restrict__ cand_cnt, int wcand,
    const double* __restrict__ lsetab_g, unsigned long long* __restrict__ queue)
{
    __shared__ double em[UN_READS * UN_EM_STRIDE];
    __shared__ double lsetab[PXG_LSE_TAB_DOUBLES];
    for (int i = threadIdx.x; i < PXG_LSE_TAB_DOUBLES; i += blockDim.x) lsetab[i] = lsetab_g[i];
    __syncthreads();
    const int lane = threadIdx.x;
    const int rr = lane >> 3, s = lane & 7;
    const int S = H.n_states;
    unsigned* bpm = bpbuf + (size_t)blockIdx.x * tmax * UN_READS;
    const int64_t n_units = unit_off[n_reads];

    // per-lane in-edge table (name-sorted slots); unused slots never win
    int src_lane[NIN];
    unsigned src_state[NIN];
    double src_lp[NIN];
#pragma unroll
    for (int d = 0; d < NIN; d++) {
        const int sidx = (s < S) ? H.in_src[s][d] : -1;
        src_state[d] = sidx >= 0 ? (unsigned)sidx : 7u;
        src_lane[d] = sidx >= 0 ? (rr * 8 + sidx) : lane;
        src_lp[d] = sidx >= 0 ? H.in_logp[s][d] : -__builtin_inf();
    }
    const double lstart = (s < S) ? H.log_start[s] : -__builtin_inf();

    // units are handed out 8 at a time from one counter: windows differ in length (the last
    // window of a read is short), and a static stride left the waves with 1 or 2 rounds each
    for (;;) {
        unsigned long long claimed = 0ull;
        if (lane == 0) claimed = atomicAdd(queue, (unsigned long long)UN_READS);
        const int64_t ubase = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(claimed >> 32)) << 32)
                                        | (unsigned)__builtin_amdgcn_readfirstlane((int)claimed));
        if (ubase >= n_units) break;
        // ---- which (read, window) is this 8-lane group's unit --------------------
        const int64_t u = ubase + rr;
        int64_t r = 0;
        bool more = u < n_units;
        const bool owned = more;
        if (more) {              // largest r with unit_off[r] <= u
            int64_t lo = 0, hi = n_reads;
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if (unit_off[mid] <= u) lo = mid; else hi = mid;
            }
            r = lo;
        }
        const UnsplitGeom g = unsplit_geometry(more ? r : n_reads, n_reads, P, cal, status, segs,
                                               first_sample, ev_off, ev_start);
        int64_t k0 = 0, k1 = -1;
        if (more) {
            const int64_t left = g.payload_start + (u - unit_off[r]) * g.window_step;
            more = g.valid && unsplit_window(g, P.stride, left, k0, k1);
        }
        int T = more ? (int)(k1 - k0 + 1) : 0;
        bool too_long = T > tmax;        // cannot happen: tmax is sized from the largest window the
        if (too_long) T = 0;             // config and the batch's sampling rates allow (host side)
        const float* x = scaled + (more ? ev_off[r] : 0);
        int Tmax = T;
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(Tmax, d);
            Tmax = o > Tmax ? o : Tmax;
        }

        // ---- forward pass ---------------------------------------------------
        // (one wave per block: LDS operations of a wave complete in order, so the phases need a
        //  compiler fence, not __syncthreads() -- whose vmcnt(0) would wait for the block means
        //  of the NEXT chunk, requested a chunk ahead)
        double v = -__builtin_inf();
        float xn[UN_CHUNK / 8];
#pragma unroll
        for (int p = 0; p < UN_CHUNK / 8; p++) {
            const int t = p * 8 + s;
            xn[p] = t < T ? x[k0 + t] : 0.0f;
        }
        for (int c0 = 0; c0 < Tmax; c0 += UN_CHUNK) {
            float xc[UN_CHUNK / 8];
#pragma unroll
            for (int p = 0; p < UN_CHUNK / 8; p++) {
                xc[p] = xn[p];
                const int t = c0 + UN_CHUNK + p * 8 + s;
                xn[p] = t < T ? x[k0 + t] : 0.0f;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int p = 0; p < UN_CHUNK / 8; p++) {
                const int tt = p * 8 + s;
                const int t = c0 + tt;
                if (t < T) {
                    const double xd = (double)xc[p];
#pragma unroll
                    for (int q = 0; q < PXG_MAX_STATES; q++)
                        if (q < S) em[rr * UN_EM_STRIDE + tt * PXG_MAX_STATES + q] = un_emission(H, lsetab, q, xd);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int tend = (Tmax - c0) < UN_CHUNK ? (Tmax - c0) : UN_CHUNK;
#pragma unroll 1
            for (int tt = 0; tt < tend; tt++) {
                const int t = c0 + tt;
                const bool act = (t < T) && (s < S);
                const double e = act ? em[rr * UN_EM_STRIDE + tt * PXG_MAX_STATES + s] : 0.0;
                double vk[NIN];
#pragma unroll
                for (int d = 0; d < NIN; d++) vk[d] = un_shfl_f64(v, src_lane[d]);   // issue all, wait once
                double best = -__builtin_inf();
                unsigned arg = 7u;
#pragma unroll
                for (int d = 0; d < NIN; d++) {
                    const double cand_v = vk[d] + src_lp[d];
                    const unsigned long long take = __ballot(cand_v > best);
                    best = pxg_sel_f64(take, best, cand_v);
                    arg = pxg_sel_u32(take, arg, src_state[d]);
                }
                const unsigned long long mact = __ballot(act);
                const double nv = (t == 0) ? (lstart + e) : (best + e);
                v = pxg_sel_f64(mact, v, nv);
                // back-pointer table of the group: source of state q at bits [3q, 3q+3)
                unsigned tbl = arg << (3 * s);
                tbl |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)tbl, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
                tbl |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)tbl, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
                tbl |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)tbl, 0x141, 0xF, 0xF, false);  // row_half_mirror
                if (s == 0) bpm[(size_t)t * UN_READS + rr] = tbl;
            }
        }
        // ---- termination ------------------------------------------------------
        double bestv = -__builtin_inf();
        int cur = H.order[0];
        for (int q = 0; q < S; q++) {
            const double vk = un_shfl_f64(v, rr * 8 + H.order[q]);
            if (q == 0 || vk > bestv) { bestv = vk; cur = H.order[q]; }
        }
        // ---- traceback + run analysis (signal_analyzer.py:393-418, backwards) --------
        int phase = 0;                   // 0 none, 1 inside an adapter run, 2 leader runs before it
        int a_last = -1, a_first = -1, lead = -1, count = 0;
        auto finalize = [&](int lead_t) {
            const int64_t ev_last = k0 + a_last, ev_lead = k0 + lead_t, ev_first = k0 + a_first;
            const int64_t adapter_end = un_ev_end(g, P.stride, ev_last);
            const int64_t leader_in_read = un_ev_start(g, P.stride, ev_lead);
            const int64_t total_duration = adapter_end - leader_in_read;
            const int64_t adapter_duration = adapter_end - un_ev_start(g, P.stride, ev_first);
            const int strict = (leader_in_read - g.payload_start) <= g.strict_duration ? 1 : 0;
            if (total_duration >= g.cut_total[strict] && adapter_duration >= g.cut_adapter[strict]) {
                if (s == 0 && count < wcand) {
                    cand[(u * wcand + count) * 2] = leader_in_read;
                    cand[(u * wcand + count) * 2 + 1] = 1 + adapter_end;
                }
                count++;
            }
        };
        __syncthreads();     // this wave's table stores (lanes s == 0) before the group's loads
        // 16 table words per round, the next round's already requested: the words do not depend
        // on the path, only the bit-field chain through them does
        constexpr int TBW = 16;
        unsigned mn[TBW];
        auto fetch = [&](int tb) {
#pragma unroll
            for (int q = 0; q < TBW; q++)
                mn[q] = (tb >= 0 && tb + q < Tmax) ? bpm[(size_t)(tb + q) * UN_READS + rr] : 0u;
        };
        const int tb_first = ((Tmax - 1) / TBW) * TBW;
        fetch(Tmax > 0 ? tb_first : -1);
        for (int tb = tb_first; tb >= 0 && Tmax > 0; tb -= TBW) {
            unsigned m[TBW];
#pragma unroll
            for (int q = 0; q < TBW; q++) m[q] = mn[q];
            fetch(tb - TBW);
#pragma unroll
            for (int q = TBW - 1; q >= 0; q--) {
                const int t = tb + q;
                if (t >= T) continue;                // also skips t >= Tmax
                const bool isA = cur == P.adapter_state;
                const bool isL = cur == P.ll_state || cur == P.lh_state;
                // a candidate closes when the runs in front of its adapter stop being leaders
                if ((phase == 1 && !isA && !isL) || (phase == 2 && !isL))
                    finalize(phase == 1 ? a_first : lead);
                if (isA && phase != 1) a_last = t;
                if (isA) a_first = t;
                if (isL && phase != 0) lead = t;
                phase = isA ? 1 : ((isL && phase != 0) ? 2 : 0);
                const int src = (int)((m[q] >> (3 * cur)) & 7u);
                if (t > 0 && src != 7) cur = src;
            }
        }
        if (phase == 1) finalize(a_first);
        else if (phase == 2) finalize(lead);
        // a window that could not be scanned, or that found more candidates than its slots hold, must not
        // pass for a clean one: both are impossible by the host-side sizing (pxg_unsplit_cand_slots, tmax) --
        // if that bound is ever wrong the read gets its own error code instead of silently losing candidates
        if (s == 0 && owned) cand_cnt[u] = (too_long || count > wcand) ? -1 : count;
    }
}

// ---- round 5: one WINDOW per lane ----------------------------------------------------------------------------------
// k_unsplit_scan above spreads the <= 8 states of a window over 8 lanes: every step fetches the in-edge values through
// the LDS crossbar (two ds_bpermute per double and edge), takes the maxima with ballot-mask selects and ORs the
// sources together over three DPP steps -- ~45 wave instructions per step for 8 windows, 70 % VALU issue and a
// quarter of the LDS cycles bank conflicts (profiles/r04/full_kernel_bounds.json).  A batch has ~6 windows per read
// (60 000 per 10 000 reads), so a LANE can own a window: its states live in its own registers, a step is
// v'[j] = max_i (v[i] + A[i][j]) + e[j] over a DENSE matrix with static register indices (a missing edge is -inf and
// never wins: strict >) -- no cross-lane traffic, no LDS on the recurrence, a step's back pointers are one coalesced
// 256-byte store.  The kernel works in RANK space (state = its position in the name-sorted order, the host permutes
// the model: UnsplitDense), so that running over the sources i = 0 .. n-1 IS the name-sorted slot order of the
// reference's argmax (worker_persistence.py:95-121 via pomegranate's edge order) and ties break as before.
// Same arithmetic, operation for operation.  For models of up to UD_S states (the reference's has 6); larger ones keep
// the 8-lane kernel.
#define UD_S 6
struct UnsplitDense {
    int n_states;
    int adapter_rank, ll_rank, lh_rank;
    int n_mix[UD_S];
    double A[UD_S][UD_S];            // log p(rank i -> rank j), -inf: no edge
    double start[UD_S];
    double mu[UD_S][PXG_MAX_MIXTURE], lssp[UD_S][PXG_MAX_MIXTURE], tss[UD_S][PXG_MAX_MIXTURE], logw[UD_S][PXG_MAX_MIXTURE];
};

// The emissions of rank q for the XC steps of a round from parameter rows in LDS ([4 kinds: mu, lssp, tss, logw][rank]
// [mixture]): the values and the operations of un_emission, written round-wide so that a rank's parameters are read
// once and the XC log-sum-exp chains of a mixture run side by side (their table reads are what a lone wave waits for).
template <int XC, int NS>
__device__ __forceinline__ void un_emission_round(const double* par, int n_mix, const double* lsetab, int q,
                                                  const float (&xc)[XC], double (&em)[XC][NS])
{
    constexpr int K = UD_S * PXG_MAX_MIXTURE;
    const double* row = par + q * PXG_MAX_MIXTURE;
    {
        const double mu = row[0], lssp = row[K], tss = row[2 * K];
#pragma unroll
        for (int j = 0; j < XC; j++) {
            const double d = (double)xc[j] - mu;
            em[j][q] = lssp - (d * d) * tss;
        }
    }
    if (n_mix > 1) {
        const double lw0 = row[3 * K];
#pragma unroll
        for (int j = 0; j < XC; j++) em[j][q] = em[j][q] + lw0;
        for (int k = 1; k < n_mix; k++) {
            const double mu = row[k], lssp = row[K + k], tss = row[2 * K + k], lw = row[3 * K + k];
#pragma unroll
            for (int j = 0; j < XC; j++) {
                const double d = (double)xc[j] - mu;
                const double l = (lssp - (d * d) * tss) + lw;
                const double a = em[j][q], b = l;
                const bool agb = a > b;
                const double m = agb ? a : b, lo = agb ? b : a;
                const double r = m + pxg_log1pexp(lsetab, lo - m);
                em[j][q] = (m == -__builtin_inf()) ? m : ((m == __builtin_inf()) ? m : r);
            }
        }
    }
}

// XC = steps per round.  4: 264 VGPRs, one wave per SIMD -- the faster kernel while a batch has no more 64-window
// groups than the chip has SIMDs (10 000 reads: 940; 2.0 ms against 2.85); 2: 234 VGPRs, two waves per SIMD, which is
// what a bigger batch needs to hide the LDS round trips (100 000 reads: 14.0 ms against 19.6).  Chosen at launch.
// EDGES (round 6): bit q * UD_S + i = the model has an edge rank i -> rank q.  The dense form adds -inf for every
// missing edge (36 add / compare / select / max groups per step); the shipped model has 16 edges, and a kernel
// specialised for ITS topology computes those alone -- a missing edge can never win and never ties (-inf > best is
// false, max(best, -inf) = best), so skipping it is the same arithmetic.  The host compares the model's topology with
// UNSPLIT_SHIPPED_EDGES and launches the specialised kernel on a match, the dense one (all bits set) otherwise
// (PXG_UNSPLIT_DENSE=1 forces it; both tested).
#define UNSPLIT_ALL_EDGES ((1ull << (UD_S * UD_S)) - 1ull)
#define UNSPLIT_SHIPPED_EDGES 0xa5027d983ull      // presets/rna-r941.cfg unsplit_read_detection_model, ranks in name order

template <int XC, unsigned long long EDGES>
__global__ __launch_bounds__(64) void k_unsplit_scan_w(
    int64_t n_reads, int tmax, UnsplitDense D, UnsplitParams P, const pxg_calib* __restrict__ cal,
    const int32_t* __restrict__ status, const int32_t* __restrict__ segs,
    const int64_t* __restrict__ first_sample, const int64_t* __restrict__ ev_off,
    const int64_t* __restrict__ ev_start, const int64_t* __restrict__ unit_off, const float* __restrict__ scaled,
    unsigned* __restrict__ bpbuf /* [wave][tmax][64 lanes] */,
    int64_t* __restrict__ cand /* n_units x wcand x 2 */, int32_t* __restrict__ cand_cnt, int wcand,
    const double* __restrict__ lsetab_g, unsigned long long* __restrict__ queue)
{
    constexpr int NS = UD_S;
    __shared__ double lsetab[PXG_LSE_TAB_DOUBLES];
    __shared__ double par[4 * NS * PXG_MAX_MIXTURE];
    __shared__ double At[NS * NS];            // [destination rank][source rank]
    for (int i = threadIdx.x; i < PXG_LSE_TAB_DOUBLES; i += blockDim.x) lsetab[i] = lsetab_g[i];
    if (threadIdx.x == 0) {
        constexpr int K = NS * PXG_MAX_MIXTURE;
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) At[j * NS + i] = D.A[i][j];
#pragma unroll
        for (int q = 0; q < NS; q++)
#pragma unroll
            for (int k = 0; k < PXG_MAX_MIXTURE; k++) {
                par[q * PXG_MAX_MIXTURE + k] = D.mu[q][k];
                par[K + q * PXG_MAX_MIXTURE + k] = D.lssp[q][k];
                par[2 * K + q * PXG_MAX_MIXTURE + k] = D.tss[q][k];
                par[3 * K + q * PXG_MAX_MIXTURE + k] = D.logw[q][k];
            }
    }
    __syncthreads();
    const int lane = threadIdx.x;
    const int S = D.n_states;
    unsigned* bpm = bpbuf + (size_t)blockIdx.x * tmax * 64 + lane;
    const int64_t n_units = unit_off[n_reads];
    // (the transition matrix is read from LDS where it is used -- one address for all lanes, a broadcast: 36 scalar
    //  pairs beside everything else get spilled into vector lanes and come back through v_readlane, 72 vector registers
    //  push the round's emissions into AGPRs and come back through v_accvgpr_read, both on the serial path)

    for (;;) {
        unsigned long long claimed = 0ull;
        if (lane == 0) claimed = atomicAdd(queue, 64ull);
        const int64_t ubase = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(claimed >> 32)) << 32)
                                        | (unsigned)__builtin_amdgcn_readfirstlane((int)claimed));
        if (ubase >= n_units) break;
        // ---- this lane's (read, window) ---------------------------------------------
        const int64_t u = ubase + lane;
        int64_t r = 0;
        bool more = u < n_units;
        const bool owned = more;
        if (more) {              // largest r with unit_off[r] <= u
            int64_t lo = 0, hi = n_reads;
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if (unit_off[mid] <= u) lo = mid; else hi = mid;
            }
            r = lo;
        }
        const UnsplitGeom g = unsplit_geometry(more ? r : n_reads, n_reads, P, cal, status, segs,
                                               first_sample, ev_off, ev_start);
        int64_t k0 = 0, k1 = -1;
        if (more) {
            const int64_t left = g.payload_start + (u - unit_off[r]) * g.window_step;
            more = g.valid && unsplit_window(g, P.stride, left, k0, k1);
        }
        int T = more ? (int)(k1 - k0 + 1) : 0;
        const bool too_long = T > tmax;  // cannot happen: tmax is sized from the largest window the
        if (too_long) T = 0;             // config and the batch's sampling rates allow (host side)
        // (a lane without a window reads the first words of the log-sum-exp table instead: a batch may have no
        //  event at all, and then there is no `scaled` to read from)
        const float* x = more ? scaled + ev_off[r] + k0 : reinterpret_cast<const float*>(lsetab_g);
        int Tmax = T;
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(Tmax, d);
            Tmax = o > Tmax ? o : Tmax;
        }
        Tmax = __builtin_amdgcn_readfirstlane(Tmax);   // (the same in every lane: scalar loop control)
        // block means are loaded unconditionally at a clamped index (a step behind a lane's window changes nothing:
        // `act`), so that the loads of the NEXT round can stay in flight while this one is worked on -- loads under
        // lane masks are waited for as a group
        const int t_last = T > 0 ? T - 1 : 0;

        // ---- forward pass: the lane's states in its own registers -------------------
        double v[NS], vfin[NS];          // vfin: the column of the lane's last step (inner rounds run on past it)
#pragma unroll
        for (int q = 0; q < NS; q++) v[q] = vfin[q] = -__builtin_inf();
        float xn[XC];
#pragma unroll
        for (int j = 0; j < XC; j++) xn[j] = x[j < t_last ? j : t_last];
        for (int c0 = 0; c0 < Tmax; c0 += XC) {
            float xc[XC];
#pragma unroll
            for (int j = 0; j < XC; j++) {
                xc[j] = xn[j];
                const int t = c0 + XC + j;
                xn[j] = x[t < t_last ? t : t_last];    // the next round's means, requested a round ahead
            }
            // emissions of the round, rank by rank (a rank's parameters are read once per round)
            double em[XC][NS];
#pragma unroll
            for (int q = 0; q < NS; q++) un_emission_round<XC, NS>(par, D.n_mix[q], lsetab, q, xc, em);
            // an INNER round: no lane's window ends in it and it is not the first -- every lane is either inside its
            // window for all XC steps or past its end (then it computes on, on means nobody reads back, into rows of its
            // own column nobody reads back; its last column is safe in vfin): no per-step selects at all
            const bool inner = c0 > 0 && __ballot(T - 1 >= c0 && T - 1 < c0 + XC) == 0ull;
            auto round = [&](auto inner_c) {
                constexpr bool INNER = decltype(inner_c)::value;
#pragma unroll
                for (int j = 0; j < XC; j++) {
                    const int t = c0 + j;
                    const bool act = t < T;                // (a round runs to its end: steps behind T change nothing)
                    asm volatile("" ::: "memory");         // At is read HERE, once per step (broadcast reads), not kept across steps
                    double nv[NS];
                    unsigned tbl = 0u;
#pragma unroll
                    for (int q = 0; q < NS; q++) {
                        double best = -__builtin_inf();
                        unsigned arg = 7u;
#pragma unroll
                        for (int i = 0; i < NS; i++) {
                            if (((EDGES >> (q * NS + i)) & 1ull) != 0ull) {        // (folded once the loops are unrolled)
                                const double cand_v = v[i] + At[q * NS + i];
                                arg = cand_v > best ? (unsigned)i : arg;          // strict: the first of equal candidates stays
                                best = __builtin_fmax(best, cand_v);               // (no NaN can arise: nothing is +inf)
                            }
                        }
                        if (INNER) {
                            nv[q] = best + em[j][q];
                        } else {
                            const double stepped = (t == 0) ? (D.start[q] + em[j][q]) : (best + em[j][q]);
                            nv[q] = act ? stepped : v[q];
                        }
                        tbl |= arg << (3 * q);
                    }
#pragma unroll
                    for (int q = 0; q < NS; q++) v[q] = nv[q];
                    if (INNER) {
                        bpm[(size_t)t * 64] = tbl;         // 64 lanes, 256 contiguous bytes (t < Tmax <= tmax rows)
                    } else {
                        if (act) bpm[(size_t)t * 64] = tbl;
#pragma unroll
                        for (int q = 0; q < NS; q++) vfin[q] = (t == T - 1) ? v[q] : vfin[q];
                    }
                }
            };
            if (inner) round(std::true_type()); else round(std::false_type());
        }
#pragma unroll
        for (int q = 0; q < NS; q++) v[q] = vfin[q];
        // ---- termination: ranks in order = the name-sorted order ---------------------
        double bestv = -__builtin_inf();
        int cur = 0;
#pragma unroll
        for (int q = 0; q < NS; q++)
            if (q < S && (q == 0 || v[q] > bestv)) { bestv = v[q]; cur = q; }
        // ---- traceback + run analysis (signal_analyzer.py:393-418, backwards) --------
        int phase = 0;                   // 0 none, 1 inside an adapter run, 2 leader runs before it
        int a_last = -1, a_first = -1, lead = -1, count = 0;
        auto finalize = [&](int lead_t) {
            const int64_t ev_last = k0 + a_last, ev_lead = k0 + lead_t, ev_first = k0 + a_first;
            const int64_t adapter_end = un_ev_end(g, P.stride, ev_last);
            const int64_t leader_in_read = un_ev_start(g, P.stride, ev_lead);
            const int64_t total_duration = adapter_end - leader_in_read;
            const int64_t adapter_duration = adapter_end - un_ev_start(g, P.stride, ev_first);
            const int strict = (leader_in_read - g.payload_start) <= g.strict_duration ? 1 : 0;
            if (total_duration >= g.cut_total[strict] && adapter_duration >= g.cut_adapter[strict]) {
                if (count < wcand) {
                    cand[(u * wcand + count) * 2] = leader_in_read;
                    cand[(u * wcand + count) * 2 + 1] = 1 + adapter_end;
                }
                count++;
            }
        };
        // (a lane reads back only the words it stored itself: program order is enough)
        constexpr int TBW = 16;
        unsigned mn[TBW];
        auto fetch = [&](int tb) {
#pragma unroll
            for (int q = 0; q < TBW; q++)
                mn[q] = (tb >= 0 && tb + q < T) ? bpm[(size_t)(tb + q) * 64] : 0u;
        };
        const int tb_first = Tmax > 0 ? ((Tmax - 1) / TBW) * TBW : 0;
        fetch(Tmax > 0 ? tb_first : -1);
        for (int tb = tb_first; tb >= 0 && Tmax > 0; tb -= TBW) {
            unsigned m[TBW];
#pragma unroll
            for (int q = 0; q < TBW; q++) m[q] = mn[q];
            fetch(tb - TBW);
#pragma unroll
            for (int q = TBW - 1; q >= 0; q--) {
                const int t = tb + q;
                if (t >= T) continue;
                const bool isA = cur == D.adapter_rank;
                const bool isL = cur == D.ll_rank || cur == D.lh_rank;
                // a candidate closes when the runs in front of its adapter stop being leaders
                if ((phase == 1 && !isA && !isL) || (phase == 2 && !isL))
                    finalize(phase == 1 ? a_first : lead);
                if (isA && phase != 1) a_last = t;
                if (isA) a_first = t;
                if (isL && phase != 0) lead = t;
                phase = isA ? 1 : ((isL && phase != 0) ? 2 : 0);
                const int src = (int)((m[q] >> (3 * cur)) & 7u);
                if (t > 0 && src != 7) cur = src;
            }
        }
        if (phase == 1) finalize(a_first);
        else if (phase == 2) finalize(lead);
        if (owned) cand_cnt[u] = (too_long || count > wcand) ? -1 : count;
    }
}

// the unsplit model in rank space (host side)
static UnsplitDense unsplit_dense(const PxgHmmDev& H, const UnsplitParams& P)
{
    UnsplitDense D;
    const int S = H.n_states;
    int rank_of[PXG_MAX_STATES];
    for (int q = 0; q < PXG_MAX_STATES; q++) rank_of[q] = -1;
    for (int i = 0; i < S; i++) rank_of[H.order[i]] = i;
    D.n_states = S;
    D.adapter_rank = P.adapter_state >= 0 && P.adapter_state < S ? rank_of[P.adapter_state] : -1;
    D.ll_rank = P.ll_state >= 0 && P.ll_state < S ? rank_of[P.ll_state] : -1;
    D.lh_rank = P.lh_state >= 0 && P.lh_state < S ? rank_of[P.lh_state] : -1;
    for (int i = 0; i < UD_S; i++) {
        D.n_mix[i] = 1;
        D.start[i] = -__builtin_inf();
        for (int j = 0; j < UD_S; j++) D.A[i][j] = -__builtin_inf();
        for (int k = 0; k < PXG_MAX_MIXTURE; k++) { D.mu[i][k] = 0.0; D.lssp[i][k] = -__builtin_inf(); D.tss[i][k] = 0.0; D.logw[i][k] = 0.0; }
    }
    for (int j = 0; j < S; j++) {
        const int q = H.order[j];
        D.n_mix[j] = H.n_mix[q];
        D.start[j] = H.log_start[q];
        for (int k = 0; k < PXG_MAX_MIXTURE; k++) { D.mu[j][k] = H.mu[q][k]; D.lssp[j][k] = H.lssp[q][k]; D.tss[j][k] = H.tss[q][k]; D.logw[j][k] = H.logw[q][k]; }
        for (int d = 0; d < PXG_MAX_STATES; d++)
            if (H.in_src[q][d] >= 0) D.A[rank_of[H.in_src[q][d]]][j] = H.in_logp[q][d];
    }
    return D;
}

// gather, pass 1: candidates per read
__global__ void k_unsplit_count(int64_t n_reads, const int64_t* __restrict__ unit_off,
                                const int32_t* __restrict__ cand_cnt, int32_t* __restrict__ out_cnt)
{
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int total = 0;
    bool failed = false;
    for (int64_t u = unit_off[r]; u < unit_off[r + 1]; u++) {
        failed |= cand_cnt[u] < 0;             // a window that overflowed its slots / its back-pointer rows
        total += cand_cnt[u] > 0 ? cand_cnt[u] : 0;
    }
    out_cnt[r] = failed ? PXG_UNSPLIT_E_WINDOW : total;
}

// gather, pass 2: the candidates of a read's windows in window order (the order the
// reference appends them), all reads back to back; iv_off = exclusive scan of out_cnt
__global__ void k_unsplit_gather(int64_t n_reads, const int64_t* __restrict__ unit_off,
                                 const int64_t* __restrict__ cand, const int32_t* __restrict__ cand_cnt, int wcand,
                                 const int32_t* __restrict__ out_cnt, const int64_t* __restrict__ iv_off,
                                 int64_t cap, int64_t* __restrict__ out_iv)
{
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= n_reads || out_cnt[r] <= 0) return;
    int64_t at = iv_off[r];
    for (int64_t u = unit_off[r]; u < unit_off[r + 1]; u++) {
        for (int q = cand_cnt[u] - 1; q >= 0; q--, at++) {   // stored last-first
            if (at < cap) {
                out_iv[at * 2] = cand[(u * wcand + q) * 2];
                out_iv[at * 2 + 1] = cand[(u * wcand + q) * 2 + 1];
            }
        }
    }
}

static UnsplitParams unsplit_params(const pxg_ctx* ctx, int stride)
{
    const pxg_config& c = ctx->cfg;
    UnsplitParams P;
    P.window_size = c.unsplit_window_size; P.window_step = c.unsplit_window_step;
    P.strict_duration = c.unsplit_strict_duration;
    P.strict_full = c.unsplit_strict_full_length; P.strict_dna = c.unsplit_strict_dna_length;
    P.loosen_full = c.unsplit_loosen_full_length; P.loosen_dna = c.unsplit_loosen_dna_length;
    P.adapter_state = c.unsplit_model.adapter_state;
    P.ll_state = c.unsplit_model.leader_low_state;
    P.lh_state = c.unsplit_model.leader_high_state;
    P.seg_adapter_state = c.segmentation_model.adapter_state;
    P.stride = stride;
    P.pool_stride = c.stride;
    return P;
}

int pxg_launch_unsplit_plan(pxg_ctx* ctx, int64_t n, const pxg_calib* cal, const int32_t* status,
                            const int32_t* segs, const int64_t* first_sample, const int64_t* ev_off,
                            int stride, int32_t* n_win, const int64_t* ev_start)
{
    if (n <= 0) return PXG_OK;
    hipLaunchKernelGGL(k_unsplit_plan, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n,
                       unsplit_params(ctx, stride), cal, status, segs, first_sample, ev_off, ev_start, n_win);
    return PXG_OK;
}

// persistent grid: enough waves to fill the chip; the unit count is only known on the device
// a model with more states than UD_S takes the 8-lane kernel; PXG_UNSPLIT_8_LANES=1 sends every model there (read at
// every call: the tests run both kernels against the oracle in one process)
static bool unsplit_lane_windows() { return !getenv("PXG_UNSPLIT_8_LANES"); }

int pxg_unsplit_waves(const pxg_ctx* ctx, int64_t units_bound)
{
    const bool lanes = unsplit_lane_windows() && ctx->hmm[1].n_states <= UD_S;
    const int per_wave = lanes ? 64 : UN_READS;
    const int64_t need = (units_bound + per_wave - 1) / per_wave;
    const int64_t cap = (int64_t)ctx->n_cu * (lanes ? 8 : 16);
    return (int)(need < cap ? (need > 0 ? need : 1) : cap);
}

size_t pxg_unsplit_scratch_bytes(const pxg_ctx* ctx, int64_t units_bound, int tmax)
{
    const bool lanes = unsplit_lane_windows() && ctx->hmm[1].n_states <= UD_S;
    return (size_t)pxg_unsplit_waves(ctx, units_bound) * (size_t)tmax * (lanes ? 64 : UN_READS) * sizeof(unsigned);
}

// most candidates a window of tmax blocks can hold (see the top of the file)
int pxg_unsplit_cand_slots(const pxg_ctx* ctx, int tmax, int stride)
{
    const pxg_config& c = ctx->cfg;
    const double shortest = std::min(c.unsplit_loosen_full_length, c.unsplit_strict_full_length);
    const int64_t min_cut = (int64_t)(shortest * ctx->rate_min);           // samples, as the kernel cuts it
    const int64_t span = (int64_t)tmax * stride + 1;
    const int64_t by_blocks = (int64_t)tmax / 2 + 2;
    const int64_t by_length = min_cut > 0 ? span / min_cut + 2 : by_blocks;
    return (int)std::max<int64_t>(1, std::min(by_blocks, by_length));
}

size_t pxg_unsplit_cand_bytes(int64_t units_bound, int wcand)
{
    return (size_t)(units_bound > 0 ? units_bound : 1) * ((size_t)wcand * 2 * sizeof(int64_t) + sizeof(int32_t));
}

int pxg_launch_unsplit_scan(pxg_ctx* ctx, int64_t n, int64_t units_bound, int tmax, const pxg_calib* cal,
                            const int32_t* status, const int32_t* segs, const int64_t* first_sample,
                            const int64_t* ev_off, const int64_t* unit_off, const float* scaled, int stride,
                            void* scratch, void* candbuf, int wcand, int32_t* out_cnt, const int64_t* ev_start)
{
    if (n <= 0) return PXG_OK;
    const UnsplitParams P = unsplit_params(ctx, stride);
    const int waves = pxg_unsplit_waves(ctx, units_bound);
    int rcq = pxg_reserve(ctx, ctx->unsplit_q, 1);
    if (rcq) return rcq;
    (void)hipMemsetAsync(ctx->unsplit_q.p, 0, sizeof(unsigned long long), ctx->stream);
    unsigned* bp = (unsigned*)scratch;
    int64_t* cand = (int64_t*)candbuf;
    int32_t* cand_cnt = (int32_t*)((char*)candbuf + (size_t)(units_bound > 0 ? units_bound : 1) * (size_t)wcand * 2 * sizeof(int64_t));
#define SCAN(K, NIN)                                                                                   \
    hipLaunchKernelGGL(K<NIN>, dim3((unsigned)waves), dim3(64), 0, ctx->stream, n,                     \
                       tmax, ctx->hmm[1], P, cal, status, segs, first_sample, ev_off, ev_start, unit_off, \
                       scaled, bp, cand, cand_cnt, wcand, ctx->d_lsetab, ctx->unsplit_q.p)
    const int nin = ctx->hmm[1].max_in;
    if (unsplit_lane_windows() && ctx->hmm[1].n_states <= UD_S) {
        const UnsplitDense D = unsplit_dense(ctx->hmm[1], P);
        // (units_bound is the host's upper bound, about 1.4 x the windows a batch really has: 10 000 reads of ~4 000
        //  blocks = 86 000 by the bound, 60 000 on the device = 940 groups for 1 024 SIMDs)
        unsigned long long edges = 0ull;            // the model's topology in rank space
        for (int i = 0; i < UD_S; i++)
            for (int q = 0; q < UD_S; q++)
                if (D.A[i][q] > -__builtin_inf()) edges |= 1ull << (q * UD_S + i);
        const bool shipped = edges == UNSPLIT_SHIPPED_EDGES && !getenv("PXG_UNSPLIT_DENSE");
        const bool four = (units_bound + 63) / 64 <= (int64_t)ctx->n_cu * 6;
#define SCANW(XC, E)                                                                                                          \
        hipLaunchKernelGGL((k_unsplit_scan_w<XC, E>), dim3((unsigned)waves), dim3(64), 0, ctx->stream, n, tmax, D, P, cal, status, \
                           segs, first_sample, ev_off, ev_start, unit_off, scaled, bp, cand, cand_cnt, wcand,                  \
                           ctx->d_lsetab, ctx->unsplit_q.p)
        if (four) { if (shipped) SCANW(4, UNSPLIT_SHIPPED_EDGES); else SCANW(4, UNSPLIT_ALL_EDGES); }
        else      { if (shipped) SCANW(2, UNSPLIT_SHIPPED_EDGES); else SCANW(2, UNSPLIT_ALL_EDGES); }
#undef SCANW
    } else {
        if (nin <= 2) SCAN(k_unsplit_scan, 2);
        else if (nin <= 3) SCAN(k_unsplit_scan, 3);
        else if (nin <= 5) SCAN(k_unsplit_scan, 5);
        else SCAN(k_unsplit_scan, 8);
    }
#undef SCAN
    hipLaunchKernelGGL(k_unsplit_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n,
                       unit_off, cand_cnt, out_cnt);
    return PXG_OK;
}

int pxg_launch_unsplit_gather(pxg_ctx* ctx, int64_t n, int64_t units_bound, const int64_t* unit_off,
                              const void* candbuf, int wcand, const int32_t* out_cnt, const int64_t* iv_off,
                              int64_t cap, int64_t* out_iv)
{
    if (n <= 0) return PXG_OK;
    const int64_t* cand = (const int64_t*)candbuf;
    const int32_t* cand_cnt = (const int32_t*)((const char*)candbuf + (size_t)(units_bound > 0 ? units_bound : 1) * (size_t)wcand * 2 * sizeof(int64_t));
    hipLaunchKernelGGL(k_unsplit_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n,
                       unit_off, cand, cand_cnt, wcand, out_cnt, iv_off, cap, out_iv);
    return PXG_OK;
}
