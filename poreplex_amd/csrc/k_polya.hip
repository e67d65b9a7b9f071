// k_polya.hip -- K6: poly(A) tail measurement (a14-a17) and the event detector
// hook (a15).  Reference: poreplex/polya.py:50-187, src/csupport.c:70-124 ->
// src/contrib/scrappie/event_detection.c:36-324.
//
// One LANE per read, PXG_PA_LANES reads per wave (the control flow is data
// dependent, so narrow waves lose less to divergence and more of them hide the
// serial latency): the peak detector is a sequential
// two-detector state machine and the retry / recalibration logic is data
// dependent, so reads are the parallel axis.  The event detector STREAMS: it
// never materialises the filtered window, prefix sums or t-statistics --
//   * 7 scaled samples slide through registers (median pre-filter, zero padded),
//   * the float64 prefix sums live in a 64-deep LDS ring per lane (the two
//     t-statistic windows need indices i-20 .. i+20 only); they are extended
//     strictly sequentially, so they are bit-identical to scrappie's loop,
//   * each detector remembers the prefix sums at its candidate peak, so an event
//     is finished the moment its right boundary is emitted.
// Events go to a wave-interleaved scratch arena ([event][lane], coalesced).
// Everything downstream (interval DP, NumPy pairwise float32 reductions, the
// open-end retry loop, recalibration, stdv QC) follows oracle/pxo_polya.c line
// by line and is bit-exact with it; the oracle is pinned by the real polya.py.
#include <float.h>
#include "pxg_common.h"

#define PA_RING 64

struct PolyaParams {
    int stride, refinement_expansion, openend_expansion, median_pre_filter, max_ext;
    int w1, w2;
    float thr1, thr2, peak_height;
    int spike_tolerance;
    double mean_loc, mean_scale, z_cutoff, stdv_max, stdv_lo, stdv_hi, spike_weight, trigger;
    int recal_max_dist, recal_min_length;
    double recal_max_stdv;
    int adapter_state, polya_state;
    int ev_cap;
};

struct Ev {            // 16 bytes, one per event and lane
    unsigned start;
    float length, mean, stdv;
};

struct Detector {
    float threshold;
    unsigned window;
    unsigned long long masked_to;
    int peak_pos;
    float peak_val;
    int valid;
    double pk_cs, pk_cq;       // prefix sums at peak_pos
};

// sample j of the inspection window: scaled full-resolution signal, 0 outside
struct WindowSrc {
    const int16_t* raw;        // RAW mode: read base
    const float* sig;          // F32 mode (event hook): window base
    int64_t ib;                // window start inside the read
    int64_t W;                 // window length
    double k, offset;
    float scale, shift;
    __device__ __forceinline__ float scaled(int64_t j) const
    {
        if (j < 0 || j >= W) return 0.0f;
        if (sig) return sig[j];
        const float pa = pxg_raw2pa(raw[ib + j], k, offset);
        const float y = scale * pa;
        return y + shift;
    }
};

__device__ __forceinline__ void cswap(float& a, float& b)
{
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo;
    b = hi;
}

__device__ __forceinline__ float median7(float a0, float a1, float a2, float a3, float a4, float a5,
                                         float a6)
{
    // 16-exchange sorting network for 7 inputs; element 3 is the median
    cswap(a1, a2); cswap(a3, a4); cswap(a5, a6);
    cswap(a0, a2); cswap(a3, a5); cswap(a4, a6);
    cswap(a0, a1); cswap(a4, a5); cswap(a2, a6);
    cswap(a0, a4); cswap(a1, a5);
    cswap(a0, a3); cswap(a2, a5);
    cswap(a1, a3); cswap(a2, a4);
    cswap(a2, a3);
    return a3;
}

// filtered sample j (median pre-filter of odd width <= 7; width 1 = none)
__device__ __forceinline__ float filtered_at(const WindowSrc& S, int64_t j, int mpf)
{
    if (mpf <= 1) return S.scaled(j);
    return median7(S.scaled(j - 3), S.scaled(j - 2), S.scaled(j - 1), S.scaled(j), S.scaled(j + 1),
                   S.scaled(j + 2), S.scaled(j + 3));
}

// t-statistic at i for window w from the prefix ring (event_detection.c:91-114)
__device__ __forceinline__ float tstat_at(const double2* ring, int lane, int64_t i, int64_t w,
                                          int64_t n)
{
    if (n < 2 * w || w < 2 || i < w || i > n - w) return 0.0f;
    const double2 c0 = ring[((i - w) & (PA_RING - 1)) * PXG_PA_LANES + lane];
    const double2 c1 = ring[(i & (PA_RING - 1)) * PXG_PA_LANES + lane];
    const double2 c2 = ring[((i + w) & (PA_RING - 1)) * PXG_PA_LANES + lane];
    const float wf = (float)w;
    double s1 = c1.x, q1 = c1.y;
    if (i > w) {
        s1 -= c0.x;
        q1 -= c0.y;
    }
    const float s2 = (float)(c2.x - c1.x);
    const float q2 = (float)(c2.y - c1.y);
    const float m1 = (float)(s1 / wf);
    const float m2 = s2 / wf;
    const float m1sq = m1 * m1;
    const float m2sq = m2 * m2;
    const float q2w = q2 / wf;
    float var = (float)(((q1 / wf - (double)m1sq) + (double)q2w) - (double)m2sq);
    var = fmaxf(var, FLT_MIN);
    const float dm = m2 - m1;
    const float vw = var / wf;
    return (float)(fabs((double)dm) / sqrt((double)vw));
}

__device__ __forceinline__ Ev make_event(unsigned long long b, unsigned long long en, double cs_b,
                                         double cq_b, double cs_e, double cq_e)
{
    Ev ev;
    ev.start = (unsigned)b;
    ev.length = (float)(en - b);
    ev.mean = (float)(cs_e - cs_b) / ev.length;
    const float dsq = (float)(cq_e - cq_b);
    const float msq = ev.mean * ev.mean;
    const float var = dsq / ev.length - msq;
    ev.stdv = sqrtf(fmaxf(var, 0.0f));
    return ev;
}

// Resumable streaming state of detect_events for one lane.
struct Stream {
    Detector det[2];
    double cs, cq;
    int64_t filled;            // prefix indices [0, filled] are in the ring
    int64_t i;                 // next sample the peak detector will visit
    float x0, x1, x2, x3, x4, x5, x6;   // scaled samples filled-3 .. filled+3
    int ne;
    unsigned long long prev_pos;
    double prev_cs, prev_cq;
};

__device__ __forceinline__ void stream_init(Stream& st, const WindowSrc& S, const PolyaParams& P,
                                            double2* ring, int lane)
{
    st.det[0] = { P.thr1, (unsigned)P.w1, 0ull, -1, FLT_MAX, 0, 0.0, 0.0 };
    st.det[1] = { P.thr2, (unsigned)P.w2, 0ull, -1, FLT_MAX, 0, 0.0, 0.0 };
    st.cs = 0.0; st.cq = 0.0; st.filled = 0; st.i = 0;
    st.x0 = st.x1 = st.x2 = 0.0f;
    st.x3 = S.scaled(0); st.x4 = S.scaled(1); st.x5 = S.scaled(2); st.x6 = S.scaled(3);
    st.ne = 0; st.prev_pos = 0; st.prev_cs = 0.0; st.prev_cq = 0.0;
    ring[0 * PXG_PA_LANES + lane] = make_double2(0.0, 0.0);
}

// Streaming detect_events over the window S (event_detection.c:273-324);
// returns the event count (>= 1); events beyond `cap` are counted, not stored.
// When `snap` is given, the state at step n - PA_SAFE (everything before it is
// independent of where the window ends: the median filter looks 3 samples
// ahead, the t-statistics 20) is saved so that an open-end retry with a longer
// window resumes there instead of starting over.
#define PA_SAFE 28
__device__ int detect_events_stream(Stream& st, const WindowSrc& S, const PolyaParams& P,
                                    double2* ring, int lane, Ev* ev /* [cap][PXG_PA_LANES] */, int cap,
                                    Stream* snap, double2* snap_ring /* [PA_RING][PXG_PA_LANES] */,
                                    int64_t* snap_n)
{
    const int64_t n = S.W;
    const int mpf = P.median_pre_filter;
    const int64_t look = P.w1 > P.w2 ? P.w1 : P.w2;    // prefix look-ahead (<= 31)
    const int64_t snap_at = n - PA_SAFE;
    for (int64_t i = st.i; i < n; i++) {
        if (snap && i == snap_at && i > 0) {
            st.i = i;
            *snap = st;
            *snap_n = n;
            for (int q = 0; q < PA_RING; q++) snap_ring[q * PXG_PA_LANES + lane] = ring[q * PXG_PA_LANES + lane];
        }
        // extend the prefix sums to index min(i + look, n), strictly in order
        const int64_t need = (i + look) < n ? (i + look) : n;
        while (st.filled < need) {
            const float f = mpf <= 1 ? st.x3 : median7(st.x0, st.x1, st.x2, st.x3, st.x4, st.x5, st.x6);
            st.x0 = st.x1; st.x1 = st.x2; st.x2 = st.x3; st.x3 = st.x4; st.x4 = st.x5; st.x5 = st.x6;
            st.x6 = S.scaled(st.filled + 4);
            const float sq = f * f;
            st.cs = st.cs + (double)f;
            st.cq = st.cq + (double)sq;
            st.filled++;
            ring[(st.filled & (PA_RING - 1)) * PXG_PA_LANES + lane] = make_double2(st.cs, st.cq);
        }
        const double2 here = ring[(i & (PA_RING - 1)) * PXG_PA_LANES + lane];
#pragma unroll
        for (int d = 0; d < 2; d++) {
            Detector& D = st.det[d];
            if (D.masked_to >= (unsigned long long)i) continue;
            const float cur = tstat_at(ring, lane, i, D.window, n);
            if (D.peak_pos == -1) {
                if (cur < D.peak_val) {
                    D.peak_val = cur;
                } else if (cur - D.peak_val > P.peak_height) {
                    D.peak_val = cur;
                    D.peak_pos = (int)i;
                    D.pk_cs = here.x;
                    D.pk_cq = here.y;
                }
            } else {
                if (cur > D.peak_val) {
                    D.peak_val = cur;
                    D.peak_pos = (int)i;
                    D.pk_cs = here.x;
                    D.pk_cq = here.y;
                }
                if (d == 0 && D.peak_val > D.threshold) {
                    st.det[1].masked_to = (unsigned long long)D.peak_pos + D.window;
                    st.det[1].peak_pos = -1;
                    st.det[1].peak_val = FLT_MAX;
                    st.det[1].valid = 0;
                }
                if (D.peak_val - cur > P.peak_height && D.peak_val > D.threshold) D.valid = 1;
                if (D.valid && ((unsigned long long)i - (unsigned long long)D.peak_pos) > D.window / 2) {
                    const unsigned long long p = (unsigned long long)D.peak_pos;
                    if (st.ne < cap)
                        ev[(size_t)st.ne * PXG_PA_LANES + lane] = make_event(st.prev_pos, p, st.prev_cs, st.prev_cq, D.pk_cs, D.pk_cq);
                    st.ne++;
                    st.prev_pos = p;
                    st.prev_cs = D.pk_cs;
                    st.prev_cq = D.pk_cq;
                    D.peak_pos = -1;
                    D.peak_val = cur;
                    D.valid = 0;
                }
            }
        }
    }
    st.i = n;
    // last event [prev, n); with no peak at all scrappie emits the single
    // zero-length event [0, peaks[0] = 0)  (event_detection.c:261-268)
    int ne = st.ne;
    if (ne < cap)
        ev[(size_t)ne * PXG_PA_LANES + lane] = ne > 0 ? make_event(st.prev_pos, (unsigned long long)n, st.prev_cs, st.prev_cq, st.cs, st.cq)
                                            : make_event(0, 0, 0.0, 0.0, 0.0, 0.0);
    return ne + 1;
}

// ---------------------------------------------------------------------------
// NumPy float32 add.reduce (pairwise) over a virtual array v(i), i in [0, n)
// ---------------------------------------------------------------------------
template <typename F>
__device__ float np_block_sum(const F& v, int64_t base, int64_t n)   // n <= 128
{
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; i++) res += v(base + i);
        return res;
    }
    float r0 = v(base), r1 = v(base + 1), r2 = v(base + 2), r3 = v(base + 3), r4 = v(base + 4),
          r5 = v(base + 5), r6 = v(base + 6), r7 = v(base + 7);
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += v(base + i); r1 += v(base + i + 1); r2 += v(base + i + 2); r3 += v(base + i + 3);
        r4 += v(base + i + 4); r5 += v(base + i + 5); r6 += v(base + i + 6); r7 += v(base + i + 7);
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += v(base + i);
    return res;
}

template <typename F>
__device__ float np_sum_f32(const F& v, int64_t n)
{
    if (n <= 128) return np_block_sum(v, 0, n);
    // explicit stack for the recursive halving (n2 = n/2 rounded down to 8)
    int64_t sb[40], sn[40];
    float acc[40];
    int st[40];       // 0: descend left, 1: left done, 2: both done
    int top = 0;
    sb[0] = 0; sn[0] = n; st[0] = 0;
    float ret = 0.0f;
    while (top >= 0) {
        const int64_t nn = sn[top];
        if (nn <= 128) {
            ret = np_block_sum(v, sb[top], nn);
            top--;
            continue;
        }
        int64_t n2 = nn / 2;
        n2 -= n2 % 8;
        if (st[top] == 0) {
            st[top] = 1;
            sb[top + 1] = sb[top]; sn[top + 1] = n2; st[top + 1] = 0;
            top++;
        } else if (st[top] == 1) {
            acc[top] = ret;
            st[top] = 2;
            sb[top + 1] = sb[top] + n2; sn[top + 1] = nn - n2; st[top + 1] = 0;
            top++;
        } else {
            ret = acc[top] + ret;
            top--;
        }
    }
    return ret;
}

// Resolve "the idx-th event in [lo, hi] that satisfies pred" for the
// monotonically increasing idx sequence np_sum_f32 produces (amortised O(1)).
struct NthCursor {
    int q, seen;
};
template <typename Pred>
__device__ __forceinline__ int nth_matching(const Pred& pred, int lo, int hi, int idx, NthCursor& c)
{
    if (idx < c.seen || c.q < lo) { c.q = lo - 1; c.seen = -1; }
    while (c.seen < idx && c.q < hi) {
        c.q++;
        if (pred(c.q)) c.seen++;
    }
    return c.q;
}

// filtered sample j through a sliding 7-sample register window: one pA
// conversion per step when j advances by one (the QC reductions do)
struct FiltCursor {
    int64_t j;
    float x0, x1, x2, x3, x4, x5, x6;
};
__device__ __forceinline__ float filtered_seq(const WindowSrc& S, int mpf, int64_t j, FiltCursor& c)
{
    if (mpf <= 1) return S.scaled(j);
    if (j == c.j + 1) {
        c.x0 = c.x1; c.x1 = c.x2; c.x2 = c.x3; c.x3 = c.x4; c.x4 = c.x5; c.x5 = c.x6;
        c.x6 = S.scaled(j + 3);
    } else if (j != c.j) {
        c.x0 = S.scaled(j - 3); c.x1 = S.scaled(j - 2); c.x2 = S.scaled(j - 1); c.x3 = S.scaled(j);
        c.x4 = S.scaled(j + 1); c.x5 = S.scaled(j + 2); c.x6 = S.scaled(j + 3);
    }
    c.j = j;
    return median7(c.x0, c.x1, c.x2, c.x3, c.x4, c.x5, c.x6);
}

// ---------------------------------------------------------------------------
// per-lane poly(A) state machine (oracle/pxo_polya.c polya_entry / call_polya /
// try_recalibrate, recursion flattened)
// ---------------------------------------------------------------------------
struct PolyaOut {
    int called, n_spikes, dwell;
    int64_t begin, end;
};

__device__ void polya_one_read(const PolyaParams& P, const int16_t* raw, int64_t n_full, double k,
                               double offset, float scale, float shift, int rough_begin,
                               int rough_end, int has_end0, double2* ring, int lane, Ev* ev,
                               double2* snap_ring, PolyaOut& out, pxg_polya_spike* spikes)
{
    out.called = 0; out.n_spikes = 0; out.dwell = 0; out.begin = 0; out.end = 0;
    const int stride = P.stride;
    const int min_unit = P.openend_expansion / stride;
    const double half = P.mean_scale * P.z_cutoff;

    int rb = rough_begin, re = rough_end, has_end = has_end0, depth = 0;
    int has_range = 0;
    double rlo = 0.0, rhi = 0.0;
    enum { ENTRY, CALL, RECAL, DONE } state = ENTRY;

    WindowSrc S;
    S.raw = raw; S.sig = nullptr; S.k = k; S.offset = offset; S.scale = scale; S.shift = shift;
    S.ib = 0; S.W = 0;
    int ne = 0;
    Stream st, snap;
    int64_t snap_n = -1;           // window length the snapshot was taken for (-1: none)
    int64_t ib = 0, ie = 0, adapter_end = 0;
    float flo = 0.0f, fhi = 0.0f;          // float32-rounded poly(A) mean range in force
    const int cap = P.ev_cap;

    auto ev_at = [&](int q) -> Ev { return ev[(size_t)q * PXG_PA_LANES + lane]; };
    auto is_polya = [&](int q) -> bool { const float m = ev_at(q).mean; return m >= flo && m <= fhi; };
    auto ev_end = [&](int q) -> int64_t {
        const Ev e = ev_at(q);
        return (int64_t)((double)e.start + (double)e.length);
    };

    for (int guard = 0; guard < 400 && state != DONE; guard++) {
        if (state == ENTRY) {
            if (!has_end || re - rb < min_unit) re = rb + min_unit;     // polya.py:55-57
            ib = (int64_t)rb * stride - P.refinement_expansion;
            if (ib < 0) ib = 0;
            ie = (int64_t)(re + 1) * stride + P.refinement_expansion;
            if (ie > n_full) ie = n_full;
            adapter_end = (int64_t)rb * stride - ib;
            if (ie - ib <= 0) { state = DONE; break; }
            S.ib = ib; S.W = ie - ib;
            if (snap_n > 0 && S.W > snap_n) {
                // open-end retry: same window start, longer window -> resume the
                // detector where the shorter window stopped being final
                st = snap;
                for (int q = 0; q < PA_RING; q++) ring[q * PXG_PA_LANES + lane] = snap_ring[q * PXG_PA_LANES + lane];
            } else {
                stream_init(st, S, P, ring, lane);
            }
            snap_n = -1;
            ne = detect_events_stream(st, S, P, ring, lane, ev, cap, &snap, snap_ring, &snap_n);
            if (ne > cap) { state = DONE; break; }         // scratch overflow: not called
            if (has_range) { flo = (float)rlo; fhi = (float)rhi; }
            else { flo = (float)(P.mean_loc - half); fhi = (float)(P.mean_loc + half); }
            state = has_end ? CALL : RECAL;
        } else if (state == CALL) {
            // ---- find_best_polya_interval (polya.py:156-187) -----------------
            // The reference fills an E x E table: cell (i,j) = sum of weights i..j
            // if the spike budget is still positive at j, else 0, and takes the
            // FIRST maximum in row-major order.  The budget only depends on the
            // trailing run of non-poly(A) events (reset to `tolerance` by every
            // poly(A) event, dead for the rest of the row once a run exceeds
            // it), so for a row that starts on a poly(A) event the valid columns
            // are the same for every row of a dead-free segment.  One backward
            // sweep with suffix sums Q[j] = sum_{k>=j} w_k therefore finds, for
            // each start i, max_j (Q[i] - Q[j+1]); ties prefer the smaller j, then
            // (across starts) the smaller i -- exactly np.argmax's first maximum.
            int64_t best = 0;
            int pi = 0, pj = -1;
            if (ne == 1) {
                const Ev e = ev_at(0);
                const bool ip = e.mean >= flo && e.mean <= fhi;
                if (ip && (int64_t)(double)e.length > 0) { best = (int64_t)(double)e.length; pi = 0; pj = 0; }
            } else {
                const int64_t tol = (int64_t)P.spike_tolerance;
                int64_t Q = 0;                 // Q[j+1] while visiting j
                int64_t minQ = 0; int minJ = -1;   // best column of the current segment
                bool have = false;
                int j = ne - 1;
                while (j >= 0) {
                    const Ev ej = ev_at(j);
                    const bool ipj = ej.mean >= flo && ej.mean <= fhi;
                    if (ipj) {
                        // column j is valid (budget = tolerance > 0)
                        if (!have || Q <= minQ) { minQ = Q; minJ = j; have = true; }
                        const int64_t wj = (int64_t)((double)ej.length);
                        const int64_t Qi = Q + wj;          // Q[j]
                        const int64_t V = Qi - minQ;         // best cell of row j
                        if (V >= best && V > 0) { best = V; pi = j; pj = minJ; }
                        Q = Qi;
                        j--;
                    } else {
                        // a run of non-poly(A) events [rs, j]: cumulative length g
                        int rs = j;
                        int64_t total = 0;
                        while (rs >= 0) {
                            const Ev er = ev_at(rs);
                            if (er.mean >= flo && er.mean <= fhi) break;
                            total += (int64_t)((double)er.length);
                            rs--;
                        }
                        rs++;
                        // walk the run backwards; g = cumulative length up to column c
                        int64_t g = total;
                        for (int c = j; c >= rs; c--) {
                            const Ev ec = ev_at(c);
                            const int64_t lc = (int64_t)((double)ec.length);
                            if (g > tol) {
                                // dead column: rows to the left cannot reach past it
                                have = false;
                            } else if (g < tol) {
                                if (!have || Q <= minQ) { minQ = Q; minJ = c; have = true; }
                            }           // g == tol: budget 0 -> cell is 0, row goes on
                            const double v = -(double)ec.length;
                            Q += (int64_t)(v * P.spike_weight);
                            g -= lc;
                        }
                        j = rs - 1;
                    }
                }
            }
            const bool found = best > 0;
            if (found && pj == ne - 1 && ie < n_full && depth < P.max_ext) {     // :81-85
                re = re + P.openend_expansion / stride;
                has_end = 1;
                depth++;
                state = ENTRY;
                continue;
            }
            bool shifted = false;
            if (found && !has_range) {                                          // :88-93
                const int m = pj - pi + 1;
                auto ml = [&](int64_t q) -> float { const Ev e = ev_at(pi + (int)q); return e.mean * e.length; };
                auto ln = [&](int64_t q) -> float { return ev_at(pi + (int)q).length; };
                const float num = np_sum_f32(ml, m);
                const float den = np_sum_f32(ln, m);
                const float level = num / den;
                shifted = fabs((double)level - P.mean_loc) > P.trigger * P.mean_scale;
            }
            if (!found || shifted) { state = RECAL; continue; }
            // ---- longest event + internal stdv QC (:101-104,150-154) ---------
            int lk = pi;
            for (int q = pi; q <= pj; q++)
                if (ev_at(q).length > ev_at(lk).length) lk = q;
            const Ev le = ev_at(lk);
            const int64_t L = (int64_t)le.length;
            const int64_t b = (int64_t)((double)le.start + (double)L * P.stdv_lo);
            const int64_t e = (int64_t)((double)le.start + (double)L * P.stdv_hi);
            bool qc_ok = false;
            if (e - b > 2) {
                const int64_t bb = b < 0 ? 0 : b, ee = e > S.W ? S.W : e;
                if (ee > bb) {
                    const int64_t cnt = ee - bb;
                    const int mpf = P.median_pre_filter;
                    FiltCursor fc;
                    fc.j = -100;
                    auto fv = [&](int64_t q) -> float { return filtered_seq(S, mpf, bb + q, fc); };
                    const float mean = np_sum_f32(fv, cnt) / (float)cnt;
                    fc.j = -100;
                    auto dv = [&](int64_t q) -> float { const float x = filtered_seq(S, mpf, bb + q, fc) - mean; return x * x; };
                    const float ss = np_sum_f32(dv, cnt);
                    const float sd = sqrtf(ss / (float)cnt);
                    qc_ok = (double)sd < P.stdv_max;
                }
            }
            if (qc_ok) {                                                        // :104-121
                const Ev e0 = ev_at(pi), e1 = ev_at(pj);
                int np_ = 0;
                // lengths of the poly(A) events, compacted in order, summed pairwise
                // (two passes over the interval: count, then a virtual array)
                for (int q = pi; q <= pj; q++) np_ += is_polya(q) ? 1 : 0;
                // virtual array of the np_ poly(A) lengths: walk with a cursor
                // (np_sum_f32 reads indices in non-decreasing blocks, but not
                // strictly sequentially) -> resolve index -> event by scan
                NthCursor pc = { pi - 1, -1 };
                auto pl = [&](int64_t idx) -> float {
                    return ev_at(nth_matching(is_polya, pi, pj, (int)idx, pc)).length;
                };
                const float dwell = np_sum_f32(pl, np_);
                int ns = 0;
                for (int q = pi; q <= pj; q++) {
                    if (!is_polya(q)) {
                        if (ns < PXG_MAX_SPIKES && spikes) {
                            spikes[ns].v[0] = ev_at(q).length;
                            spikes[ns].v[1] = q - 1 >= pi ? ev_at(q - 1).mean : __builtin_nanf("");
                            spikes[ns].v[2] = ev_at(q).mean;
                            spikes[ns].v[3] = q + 1 <= pj ? ev_at(q + 1).mean : __builtin_nanf("");
                        }
                        ns++;
                    }
                }
                out.called = 1;
                out.begin = (int64_t)e0.start + ib;
                out.end = (int64_t)((double)e1.start + (double)e1.length) + ib;
                out.dwell = (int)dwell;
                out.n_spikes = ns > 127 ? 127 : ns;
                state = DONE;
                continue;
            }
            state = has_range ? DONE : RECAL;                                   // :122-125
        } else {   // RECAL: try_recalibrate_shifted_signal (polya.py:127-148)
            int m = 0;
            auto anchor = [&](int q) -> bool {
                const Ev e = ev_at(q);
                return (int64_t)e.start <= adapter_end + P.recal_max_dist && ev_end(q) > adapter_end &&
                       (double)e.stdv < P.recal_max_stdv;
            };
            for (int q = 0; q < ne; q++) m += anchor(q) ? 1 : 0;
            if (m == 0) { state = DONE; continue; }
            NthCursor ac = { -1, -1 };
            auto aml = [&](int64_t idx) -> float {
                const Ev e = ev_at(nth_matching(anchor, 0, ne - 1, (int)idx, ac));
                return e.mean * e.length;
            };
            const float asum = np_sum_f32(aml, m);
            ac = { -1, -1 };
            auto aln = [&](int64_t idx) -> float {
                return ev_at(nth_matching(anchor, 0, ne - 1, (int)idx, ac)).length;
            };
            const float pm = asum / np_sum_f32(aln, m);
            rlo = (double)pm - half;
            rhi = (double)pm + half;
            flo = (float)rlo;
            fhi = (float)rhi;
            int np_ = 0;
            for (int q = 0; q < ne; q++) np_ += is_polya(q) ? 1 : 0;
            NthCursor pc = { -1, -1 };
            auto pl = [&](int64_t idx) -> float {
                return ev_at(nth_matching(is_polya, 0, ne - 1, (int)idx, pc)).length;
            };
            const float tot = np_sum_f32(pl, np_);
            if ((double)tot >= (double)P.recal_min_length) {
                has_range = 1;
                state = CALL;
            } else {
                state = DONE;
            }
        }
    }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_polya(int64_t n_reads, PolyaParams P,
                                              const int16_t* __restrict__ raw,
                                              const int64_t* __restrict__ off,
                                              const pxg_calib* __restrict__ cal,
                                              const float* __restrict__ ss,
                                              const int32_t* __restrict__ status,
                                              const int32_t* __restrict__ segs, Ev* __restrict__ evbuf,
                                              double2* __restrict__ snapbuf,
                                              int32_t* __restrict__ pout /* n x 8 */,
                                              pxg_polya_spike* __restrict__ spikes)
{
    __shared__ double2 ring[PA_RING * PXG_PA_LANES];
    const int lane = threadIdx.x;
    const int64_t r = blockIdx.x * (int64_t)PXG_PA_LANES + lane;
    if (lane >= PXG_PA_LANES || r >= n_reads) return;
    int32_t* po = pout + r * 8;
    for (int q = 0; q < 8; q++) po[q] = 0;
    pxg_polya_spike* sp = spikes + r * PXG_MAX_SPIKES;
    if (status[r] != PXG_ST_OKAY) return;
    const int32_t* first = segs + r * 2 * PXG_N_SEGMENTS;
    const int32_t* last = first + PXG_N_SEGMENTS;
    if (P.adapter_state < 0 || first[P.adapter_state] < 0) return;
    int rb, re, has_end;
    if (P.polya_state >= 0 && first[P.polya_state] >= 0) {         // signal_analyzer.py:251-256
        rb = first[P.polya_state]; re = last[P.polya_state]; has_end = 1;
    } else {
        rb = last[P.adapter_state] + 1; re = 0; has_end = 0;
    }
    const pxg_calib c = cal[r];
    PolyaOut out;
    Ev* ev = evbuf + (size_t)blockIdx.x * P.ev_cap * PXG_PA_LANES;
    double2* snap_ring = snapbuf + (size_t)blockIdx.x * PA_RING * PXG_PA_LANES;
    polya_one_read(P, raw + off[r], off[r + 1] - off[r], c.range / c.digitisation, c.offset,
                   ss[2 * r], ss[2 * r + 1], rb, re, has_end, ring, lane, ev, snap_ring, out, sp);
    po[0] = out.called;
    po[1] = out.n_spikes;
    po[2] = out.dwell;
    po[3] = (int32_t)(out.begin & 0xFFFFFFFFll);
    po[4] = (int32_t)(out.begin >> 32);
    po[5] = (int32_t)(out.end & 0xFFFFFFFFll);
    po[6] = (int32_t)(out.end >> 32);
}

// a15 hook: detect_events on caller-supplied float windows (csupport.detect_events)
__global__ __launch_bounds__(64) void k_detect_events(int64_t n_windows, PolyaParams P,
                                                      const float* __restrict__ sig,
                                                      const int64_t* __restrict__ off,
                                                      Ev* __restrict__ evbuf,
                                                      int64_t* __restrict__ n_events)
{
    __shared__ double2 ring[PA_RING * PXG_PA_LANES];
    const int lane = threadIdx.x;
    const int64_t r = blockIdx.x * (int64_t)PXG_PA_LANES + lane;
    if (lane >= PXG_PA_LANES || r >= n_windows) return;
    WindowSrc S;
    S.raw = nullptr; S.sig = sig + off[r]; S.ib = 0; S.W = off[r + 1] - off[r];
    S.k = 0; S.offset = 0; S.scale = 1; S.shift = 0;
    if (S.W <= 0) { n_events[r] = 0; return; }
    Ev* ev = evbuf + (size_t)blockIdx.x * P.ev_cap * PXG_PA_LANES;
    Stream st;
    stream_init(st, S, P, ring, lane);
    n_events[r] = detect_events_stream(st, S, P, ring, lane, ev, P.ev_cap, nullptr, nullptr, nullptr);
}

static PolyaParams make_params(const pxg_config& c, int ev_cap, int mpf)
{
    PolyaParams P;
    P.stride = c.stride;
    P.refinement_expansion = c.polya_refinement_expansion;
    P.openend_expansion = c.polya_openend_expansion;
    P.median_pre_filter = mpf;
    P.max_ext = c.polya_maximum_openend_extension;
    P.w1 = c.ed_window_length1; P.w2 = c.ed_window_length2;
    P.thr1 = c.ed_threshold1; P.thr2 = c.ed_threshold2; P.peak_height = c.ed_peak_height;
    P.spike_tolerance = c.polya_spike_tolerance;
    P.mean_loc = c.polya_mean_dist[0]; P.mean_scale = c.polya_mean_dist[1];
    P.z_cutoff = c.polya_mean_z_cutoff; P.stdv_max = c.polya_stdv_max;
    P.stdv_lo = c.polya_stdv_range[0]; P.stdv_hi = c.polya_stdv_range[1];
    P.spike_weight = c.polya_spike_weight; P.trigger = c.polya_mean_trigger_recalibration;
    P.recal_max_dist = c.recal_max_dist_from_adapter; P.recal_min_length = c.recal_min_length;
    P.recal_max_stdv = c.recal_max_stdv;
    P.adapter_state = c.segmentation_model.adapter_state;
    P.polya_state = c.segmentation_model.polya_state;
    P.ev_cap = ev_cap;
    return P;
}

int pxg_polya_supported(pxg_ctx* ctx)
{
    const pxg_config& c = ctx->cfg;
    if ((c.polya_median_pre_filter != 7 && c.polya_median_pre_filter > 1) ||
        c.ed_window_length1 > 31 || c.ed_window_length2 > 31 || c.ed_window_length1 < 1 ||
        c.ed_window_length2 < 1) {
        ctx->err = "poly(A) kernel supports median_pre_filter in {<=1, 7} and event windows <= 31";
        return PXG_E_UNSUPPORTED;
    }
    return PXG_OK;
}

#define PA_EV_CAP 4096

int pxg_launch_polya(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                     const pxg_calib* cal, const float* ss, const int32_t* status,
                     const int32_t* segs, int32_t* pout, pxg_polya_spike* spikes)
{
    if (n <= 0) return PXG_OK;
    int rc = pxg_polya_supported(ctx);
    if (rc) return rc;
    const int64_t blocks = (n + PXG_PA_LANES - 1) / PXG_PA_LANES;
    const size_t ev_bytes = (size_t)blocks * PA_EV_CAP * PXG_PA_LANES * sizeof(Ev);
    const size_t snap_bytes = (size_t)blocks * PA_RING * PXG_PA_LANES * sizeof(double2);
    if ((rc = pxg_reserve(ctx, ctx->polya_ev, ev_bytes + snap_bytes))) return rc;
    const PolyaParams P = make_params(ctx->cfg, PA_EV_CAP, ctx->cfg.polya_median_pre_filter);
    hipLaunchKernelGGL(k_polya, dim3((unsigned)blocks), dim3(64), 0, ctx->stream, n, P, raw, off, cal,
                       ss, status, segs, (Ev*)ctx->polya_ev.p, (double2*)(ctx->polya_ev.p + ev_bytes),
                       pout, spikes);
    return PXG_OK;
}

int pxg_launch_detect_events(pxg_ctx* ctx, int64_t n, const float* sig, const int64_t* off,
                             int64_t cap, void* evbuf, int64_t* n_events)
{
    if (n <= 0) return PXG_OK;
    int rc = pxg_polya_supported(ctx);
    if (rc) return rc;
    const PolyaParams P = make_params(ctx->cfg, (int)cap, 1);     // hook: no pre-filter
    hipLaunchKernelGGL(k_detect_events, dim3((unsigned)((n + PXG_PA_LANES - 1) / PXG_PA_LANES)), dim3(64), 0, ctx->stream, n, P,
                       sig, off, (Ev*)evbuf, n_events);
    return PXG_OK;
}
