// k_polya.hip -- K6: poly(A) tail measurement (a14-a17) and the event detector
// hook (a15).  Reference: poreplex/polya.py:50-187, src/csupport.c:70-124 ->
// src/contrib/scrappie/event_detection.c:36-324.
//
// 16 lanes per read, PXG_PA_LANES = 4 reads per wave.  Measured on the way here
// (profiles/r01): with one lane per read the batch ran on 157 waves
// (15 % of the SIMDs) and the per-sample cost -- six f64 divisions and two f64
// square roots in the two t-statistics -- sat on a serial path; the data-dependent
// FSM / retry / recalibration code diverges almost completely between reads, so
// its cost grows with the reads per wave (16 reads/wave: 13.2 ms, 8: 9.1, 4: 7.4,
// 2: 8.5, 1: 8.5-9.0).  So: the independent work (median pre-filter, t-statistics,
// event-row and filtered-sample fetches) runs 16 lanes wide per read, everything
// order-dependent (float64 prefix sums, peak FSM, interval search, NumPy pairwise
// reductions, retry loop) is group-uniform: all 16 lanes of a read execute it with
// identical values, lane 0 stores.
//   * the float64 prefix sums live in a 128-deep LDS ring per read and are extended
//     strictly sequentially, so they are bit-identical to scrappie's loop,
//   * each detector remembers the prefix sums at its candidate peak, so an event
//     is finished the moment its right boundary is emitted,
//   * open-end retries (polya.py:81-85) resume from a snapshot of the detector
//     taken at the last chunk that cannot see the window end.
// Events go to a wave-interleaved scratch arena ([event][read slot]).
// Everything downstream (interval DP, NumPy pairwise float32 reductions, the
// open-end retry loop, recalibration, stdv QC) follows oracle/pxo_polya.c line
// by line and is bit-exact with it; the oracle is pinned by the real polya.py.
#include <float.h>
#include <algorithm>
#include <string.h>
#include "pxg_common.h"


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
    // two-phase form of scaled(): fetch() only issues the load (so a caller can keep it in
    // flight across other work), finish() converts; finish(fetch(j), j) == scaled(j)
    __device__ __forceinline__ float fetch(int64_t j) const
    {
        if (j < 0 || j >= W) return 0.0f;
        return sig ? sig[j] : (float)raw[ib + j];          // int16 -> float is exact
    }
    __device__ __forceinline__ float finish(float v, int64_t j) const
    {
        if (j < 0 || j >= W) return 0.0f;
        if (sig) return v;
        const float pa = (float)(k * ((double)v + offset));
        const float y = scale * pa;
        return y + shift;
    }
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

// Per-read LDS workspace of the cooperative event detector (16 lanes per read)
#ifndef PA_WAVES_PER_EU
#define PA_WAVES_PER_EU 3        // 4 reads per wave: 10 000 reads = 2500 waves must be co-resident
#endif
#define PA_GL (64 / PXG_PA_LANES) // lanes per read
#ifndef PA_EVC
#define PA_EVC 16                // event rows in the LDS chunk cache
#endif
#define PA_XS 64                 // scaled-sample ring (needs filled-3 .. filled+2*PA_GL+3)
#define PA_OVER_HEAD 4            // words in front of the read ids of the overflow list (polya_over)
#define PA_PRE 128               // prefix-sum ring (needs i-31 .. i+31+16 around a 16-sample chunk)
struct GroupLds {
    float fbuf[PA_GL];           // filtered samples of the chunk being accumulated (detector) / summed (reductions)
    // The detector's rings and what runs behind the detector are never live at the same time (detect_events_group
    // leaves nothing in LDS that polya_one_read reads afterwards: the event rows are in HBM, its resume point in
    // registers and HBM; ev_at's chunk cache is dropped after every detector call).  3 168 bytes per read instead of
    // 3 648: twelve 4-read waves per CU instead of ten, 3 072 blocks per launch round instead of 2 560.
    union {
        struct {                 // ---- detect_events_group
            double2 pre[PA_PRE];             // pre[k & 127] = (sum, sum of squares) of filtered samples [0, k)
            float tb[2][PA_GL];              // t-statistics of the chunk being scanned
            float xs[PA_XS];                 // scaled samples ring: xs[k & 63] = sample k of the window
            double2 bsum[2 * PA_GL + 1];     // event boundaries found in the chunk being scanned: prefix sums ...
            unsigned bpos[2 * PA_GL + 1];    // ... and positions; [0] = the last one before the chunk
        };
        struct {                 // ---- everything behind it
            Ev evc[PA_EVC];                  // consecutive event rows (chunk cache of the event passes)
            float vb[128];                   // reductions: one leaf block of a NumPy pairwise sum + its
            int sk_base[24], sk_n[24], sk_st[24];    // recursion stack (depth <= log2(n / 64))
            float sk_acc[24];
        };
    };
};
static_assert(sizeof(GroupLds) <= 3200, "GroupLds grew: 12 waves per CU need <= 3 328 bytes per read");

// Resume point of the detector for open-end retries (polya.py:81-85: same window
// start, longer window): the group-uniform registers at the start of the last
// chunk whose inputs cannot see the window end; the prefix ring goes to HBM.
struct GroupSnap {
    int64_t ib, n, i0, filled;   // n: window length it was taken for (-1: none)
    Detector det[2];
    double cs, cq, prev_cs, prev_cq;
    unsigned long long prev_pos;
    int ne;
};

// t-statistic at i for window w from the prefix ring (event_detection.c:91-114)
__device__ __forceinline__ float tstat_at(const double2* pre, int64_t i, int64_t w, int64_t n)
{
    if (n < 2 * w || w < 2 || i < w || i > n - w) return 0.0f;
    const double2 c0 = pre[(i - w) & (PA_PRE - 1)];
    const double2 c1 = pre[i & (PA_PRE - 1)];
    const double2 c2 = pre[(i + w) & (PA_PRE - 1)];
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

// detect_events over the window S (event_detection.c:273-324), one read per 16
// lanes.  Per 16-sample chunk: (A) the lanes filter 16 samples in parallel, (B)
// the float64 prefix sums are extended strictly in order (every lane of the
// group runs the same 16 adds, so they stay bit-identical to scrappie's loop),
// (C) the lanes evaluate the two t-statistics of 16 samples in parallel -- six
// f64 divisions and two f64 square roots per sample, the bulk of the work --
// (D) the two-detector peak FSM walks the 16 results.  Everything outside (A)
// and (C) is group-uniform: all 16 lanes hold the same state.  Returns the event
// count (>= 1); events beyond `cap` are counted, not stored.
__device__ int detect_events_group(const WindowSrc& S, const PolyaParams& P, GroupLds* L, int gl, int grp,
                                   Ev* ev /* [cap][PXG_PA_LANES] */, int cap,
                                   GroupSnap* snap /* in: resume point or n < 0; out: new one */,
                                   double2* snap_ring /* [PA_PRE], this read's */)
{
    const int64_t n = S.W;
    const int mpf = P.median_pre_filter;
    const int64_t look = P.w1 > P.w2 ? P.w1 : P.w2;    // prefix look-ahead (<= 31)
    Detector det[2];
    det[0] = { P.thr1, (unsigned)P.w1, 0ull, -1, FLT_MAX, 0, 0.0, 0.0 };
    det[1] = { P.thr2, (unsigned)P.w2, 0ull, -1, FLT_MAX, 0, 0.0, 0.0 };
    double cs = 0.0, cq = 0.0;
    int64_t filled = 0;
    int ne = 0;
    unsigned long long prev_pos = 0;
    double prev_cs = 0.0, prev_cq = 0.0;
    int64_t i_begin = 0;
    if (snap && snap->n > 0 && snap->ib == S.ib && n > snap->n) {
        // resume: everything before chunk snap->i0 is identical for the longer window
        det[0] = snap->det[0]; det[1] = snap->det[1];
        cs = snap->cs; cq = snap->cq; filled = snap->filled; ne = snap->ne;
        prev_pos = snap->prev_pos; prev_cs = snap->prev_cs; prev_cq = snap->prev_cq;
        i_begin = snap->i0;
        for (int q = gl; q < PA_PRE; q += PA_GL) L->pre[q] = snap_ring[q];
    } else if (gl == 0) {
        L->pre[0] = make_double2(0.0, 0.0);
    }
    if (snap) snap->n = -1;
    // sample ring: restart it 3 samples before the first prefix still to be extended (0 outside
    // the window, like scipy's medfilt padding); slots of negative indices read as zero
    int64_t xfilled = filled - 3;
    float xraw = S.fetch(xfilled + gl);
    // last chunk start whose t-statistics (look-ahead <= 31), prefix sums (up to 15 more)
    // and median filter (3 more) all stay inside this window
    const int64_t snap_i0 = ((n - 51) / PA_GL) * PA_GL;
    __builtin_amdgcn_wave_barrier();

    for (int64_t i0 = i_begin; i0 < n; i0 += PA_GL) {
        if (snap && i0 == snap_i0 && i0 > i_begin) {
            snap->ib = S.ib; snap->n = n; snap->i0 = i0; snap->filled = filled;
            snap->det[0] = det[0]; snap->det[1] = det[1];
            snap->cs = cs; snap->cq = cq; snap->ne = ne;
            snap->prev_pos = prev_pos; snap->prev_cs = prev_cs; snap->prev_cq = prev_cq;
            for (int q = gl; q < PA_PRE; q += PA_GL) snap_ring[q] = L->pre[q];
        }
        // ---- (A)+(B): prefix sums up to index min(i0 + 15 + look, n) -----------------
        const int64_t last = i0 + PA_GL - 1 + look;
        const int64_t need = last < n ? last : n;
        while (filled < need) {
            // scaled samples up to filled + PA_GL + 3 into the ring; every lane converts ONE
            // sample per step and its successor's load is already in flight
            while (xfilled < filled + PA_GL + 3) {
                L->xs[(xfilled + gl) & (PA_XS - 1)] = S.finish(xraw, xfilled + gl);
                xfilled += PA_GL;
                xraw = S.fetch(xfilled + gl);
            }
            __builtin_amdgcn_wave_barrier();
            const int64_t j = filled + gl;
            float f = 0.0f;
            if (j < n) {
                const float* X = L->xs;
                f = mpf <= 1 ? X[j & (PA_XS - 1)]
                             : median7(X[(j - 3) & (PA_XS - 1)], X[(j - 2) & (PA_XS - 1)],
                                       X[(j - 1) & (PA_XS - 1)], X[j & (PA_XS - 1)],
                                       X[(j + 1) & (PA_XS - 1)], X[(j + 2) & (PA_XS - 1)],
                                       X[(j + 3) & (PA_XS - 1)]);
            }
            L->fbuf[gl] = f;
            __builtin_amdgcn_wave_barrier();
            const int64_t m = (n - filled) < PA_GL ? (n - filled) : PA_GL;
#pragma unroll
            for (int q = 0; q < PA_GL; q++) {
                if (q < m) {
                    const float fq = L->fbuf[q];
                    const float sq = fq * fq;
                    cs = cs + (double)fq;
                    cq = cq + (double)sq;
                    if (gl == 0) L->pre[(filled + 1 + q) & (PA_PRE - 1)] = make_double2(cs, cq);
                }
            }
            filled += m;
            __builtin_amdgcn_wave_barrier();
        }
        // ---- (C): the two t-statistics of sample i0 + gl ---------------------------------
        {
            const int64_t i = i0 + gl;
            float t1 = 0.0f, t2 = 0.0f;
            if (i < n) {
                t1 = tstat_at(L->pre, i, det[0].window, n);
                t2 = tstat_at(L->pre, i, det[1].window, n);
            }
            L->tb[0][gl] = t1;
            L->tb[1][gl] = t2;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- (D): peak FSM over the chunk (group-uniform) ----------------------------------
        // A boundary only costs an LDS append here; the event rows (two float divisions and
        // a square root each) are materialised afterwards, one lane per event.
        const int jn = (n - i0) < PA_GL ? (int)(n - i0) : PA_GL;
        const int ne0 = ne;
        int nb = 0;
        if (gl == 0) {
            L->bsum[0] = make_double2(prev_cs, prev_cq);
            L->bpos[0] = (unsigned)prev_pos;
        }
        // next sample's operands are requested one iteration ahead (LDS latency off the chain)
        double2 here_n = L->pre[i0 & (PA_PRE - 1)];
        float tn0 = L->tb[0][0], tn1 = L->tb[1][0];
#pragma unroll 1
        for (int jj = 0; jj < jn; jj++) {
            const int64_t i = i0 + jj;
            const double2 here = here_n;
            const float tcur[2] = { tn0, tn1 };
            if (jj + 1 < jn) {
                here_n = L->pre[(i + 1) & (PA_PRE - 1)];
                tn0 = L->tb[0][jj + 1];
                tn1 = L->tb[1][jj + 1];
            }
#pragma unroll
            for (int d = 0; d < 2; d++) {
                Detector& D = det[d];
                if (D.masked_to >= (unsigned long long)i) continue;
                const float cur = tcur[d];
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
                        det[1].masked_to = (unsigned long long)D.peak_pos + D.window;
                        det[1].peak_pos = -1;
                        det[1].peak_val = FLT_MAX;
                        det[1].valid = 0;
                    }
                    if (D.peak_val - cur > P.peak_height && D.peak_val > D.threshold) D.valid = 1;
                    if (D.valid && ((unsigned long long)i - (unsigned long long)D.peak_pos) > D.window / 2) {
                        // a boundary: (position, prefix sums) appended to the chunk's list
                        nb++;
                        if (gl == 0) {
                            L->bsum[nb] = make_double2(D.pk_cs, D.pk_cq);
                            L->bpos[nb] = (unsigned)D.peak_pos;
                        }
                        ne++;
                        prev_pos = (unsigned long long)D.peak_pos;
                        prev_cs = D.pk_cs;
                        prev_cq = D.pk_cq;
                        D.peak_pos = -1;
                        D.peak_val = cur;
                        D.valid = 0;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = gl; k < nb; k += PA_GL) {           // event k of the chunk = [boundary k, k + 1)
            if (ne0 + k < cap) {
                const double2 b0 = L->bsum[k], b1 = L->bsum[k + 1];
                ev[(size_t)(ne0 + k) * PXG_PA_LANES + grp] =
                    make_event(L->bpos[k], L->bpos[k + 1], b0.x, b0.y, b1.x, b1.y);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // last event [prev, n); with no peak at all scrappie emits the single
    // zero-length event [0, peaks[0] = 0)  (event_detection.c:261-268)
    if (ne < cap && gl == 0)
        ev[(size_t)ne * PXG_PA_LANES + grp] = ne > 0 ? make_event(prev_pos, (unsigned long long)n, prev_cs, prev_cq, cs, cq)
                                                    : make_event(0, 0, 0.0, 0.0, 0.0, 0.0);
    __threadfence_block();        // the group's event rows before any lane of it reads them
    return ne + 1;
}

// ---------------------------------------------------------------------------
// NumPy float32 add.reduce (pairwise) over a virtual array v(i), i in [0, n)
// ---------------------------------------------------------------------------
// The virtual array is consumed strictly in index order, one leaf block (<= 128
// elements) at a time: the block is first staged into LDS by ONE loop around v() --
// a single inlined copy of the caller's lambda instead of 18, which kept this kernel
// above the instruction cache -- and then summed by plain code; the recursion stack
// of the pairwise split lives in LDS as well (it used to be private memory: 3.2 of
// the 6.6 ms of the whole poly(A) stage).  All of it is group-uniform.
__device__ float np_block_sum_lds(const float* vb, int n)   // n <= 128
{
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; i++) res += vb[i];
        return res;
    }
    float r0 = vb[0], r1 = vb[1], r2 = vb[2], r3 = vb[3], r4 = vb[4], r5 = vb[5], r6 = vb[6], r7 = vb[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += vb[i]; r1 += vb[i + 1]; r2 += vb[i + 2]; r3 += vb[i + 3];
        r4 += vb[i + 4]; r5 += vb[i + 5]; r6 += vb[i + 6]; r7 += vb[i + 7];
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += vb[i];
    return res;
}

template <typename F>
__device__ __forceinline__ float np_leaf(const F& v, int64_t base, int n, GroupLds* L, int gl)
{
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        const float x = v(base + i);
        if (gl == 0) L->vb[i] = x;
    }
    __builtin_amdgcn_wave_barrier();
    return np_block_sum_lds(L->vb, n);
}

template <typename F>
__device__ float np_sum_f32(const F& v, int64_t n, GroupLds* L, int gl)
{
    if (n <= 128) return np_leaf(v, 0, (int)n, L, gl);
    // explicit stack for the recursive halving (n2 = n/2 rounded down to 8), in LDS
    int top = 0;
    if (gl == 0) { L->sk_base[0] = 0; L->sk_n[0] = (int)n; L->sk_st[0] = 0; }
    float ret = 0.0f;
    while (top >= 0) {
        __builtin_amdgcn_wave_barrier();
        const int nn = L->sk_n[top], bb = L->sk_base[top], st = L->sk_st[top];
        if (nn <= 128) {
            ret = np_leaf(v, bb, nn, L, gl);
            top--;
            continue;
        }
        int n2 = nn / 2;
        n2 -= n2 % 8;
        if (st == 0) {
            if (gl == 0) {
                L->sk_st[top] = 1;
                L->sk_base[top + 1] = bb; L->sk_n[top + 1] = n2; L->sk_st[top + 1] = 0;
            }
            top++;
        } else if (st == 1) {
            if (gl == 0) {
                L->sk_acc[top] = ret;
                L->sk_st[top] = 2;
                L->sk_base[top + 1] = bb + n2; L->sk_n[top + 1] = nn - n2; L->sk_st[top + 1] = 0;
            }
            top++;
        } else {
            ret = L->sk_acc[top] + ret;
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

// filtered sample j for the group-uniform reductions: the 16 lanes of a read
// filter one aligned 16-sample chunk in parallel into LDS, the (sequential,
// NumPy-ordered) reduction then reads it back; np_sum_f32 visits indices in
// increasing order, so every chunk is built once.
struct FiltCache {
    int64_t c0;                  // first sample of the chunk held in L->fbuf, -1: none
};
__device__ __forceinline__ float filtered_seq(const WindowSrc& S, int mpf, int64_t j, FiltCache& c,
                                              GroupLds* L, int gl)
{
    const int64_t cb = j & ~(int64_t)(PA_GL - 1);
    if (cb != c.c0) {                      // group-uniform
        __builtin_amdgcn_wave_barrier();
        L->fbuf[gl] = filtered_at(S, cb + gl, mpf);
        __builtin_amdgcn_wave_barrier();
        c.c0 = cb;
    }
    return L->fbuf[j & (PA_GL - 1)];
}

// ---------------------------------------------------------------------------
// per-lane poly(A) state machine (oracle/pxo_polya.c polya_entry / call_polya /
// try_recalibrate, recursion flattened)
// ---------------------------------------------------------------------------
struct PolyaOut {
    int called, n_spikes, dwell;
    int64_t begin, end;
    int overflow;          // events the window needed when the scratch rows ran out (0 = fitted)
    int spike_base;        // first row of this read's spikes in the batch's spike arena
    int spill;             // the arena had no room left for them: the read is listed and re-run
};

// Spike rows of all reads share one arena; a called tail takes its rows with one atomic add
// (every spike is kept: polya.py:109-115 lists them all -- no per-read limit).
struct SpikeSink {
    pxg_polya_spike* arena;    // null: count only
    int cap;                   // rows in the arena
    int32_t* cursor;           // rows handed out so far (may run past cap: that is the demand)
};

__device__ void polya_one_read(const PolyaParams& P, const int16_t* raw, int64_t n_full, double k,
                               double offset, float scale, float shift, int rough_begin,
                               int rough_end, int has_end0, GroupLds* L, int gl, int lane /* read slot of the wave */,
                               Ev* ev, double2* snap_ring, PolyaOut& out, SpikeSink sink)
{
    out.called = 0; out.n_spikes = 0; out.dwell = 0; out.begin = 0; out.end = 0; out.overflow = 0;
    out.spike_base = 0; out.spill = 0;
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
    GroupSnap snap;
    snap.n = -1;
    int64_t ib = 0, ie = 0, adapter_end = 0;
    float flo = 0.0f, fhi = 0.0f;          // float32-rounded poly(A) mean range in force
    const int cap = P.ev_cap;

    // event rows through a 16-row LDS chunk (one coalesced fetch by the 16 lanes
    // instead of a dependent global load per event and pass)
    int evc0 = -1;
    auto ev_at = [&](int q) -> Ev {
        const int cb = q & ~(PA_EVC - 1);
        if (cb != evc0) {                       // group-uniform
            __builtin_amdgcn_wave_barrier();
            for (int k = gl; k < PA_EVC; k += PA_GL) L->evc[k] = ev[(size_t)(cb + k) * PXG_PA_LANES + lane];
            __builtin_amdgcn_wave_barrier();
            evc0 = cb;
        }
        return L->evc[q & (PA_EVC - 1)];
    };
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
            ne = detect_events_group(S, P, L, gl, lane, ev, cap, &snap, snap_ring);
            evc0 = -1;
            if (ne > cap) { out.overflow = ne; state = DONE; break; }   // scratch rows ran out: the read is queued for the
                                                                        // retry pass with more rows (pxg_polya_settle)
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
                const float num = np_sum_f32(ml, m, L, gl);
                const float den = np_sum_f32(ln, m, L, gl);
                const float level = num / den;
                shifted = fabs((double)level - P.mean_loc) > P.trigger * P.mean_scale;
            }
            if (!found || shifted) { state = RECAL; continue; }
            // ---- longest event + internal stdv QC (:101-104,150-154) ---------
            int lk = pi;
            for (int q = pi; q <= pj; q++)
                if (ev_at(q).length > ev_at(lk).length) lk = q;
            const Ev le = ev_at(lk);
            const int64_t Llen = (int64_t)le.length;
            const int64_t b = (int64_t)((double)le.start + (double)Llen * P.stdv_lo);
            const int64_t e = (int64_t)((double)le.start + (double)Llen * P.stdv_hi);
            bool qc_ok = false;
            if (e - b > 2) {
                const int64_t bb = b < 0 ? 0 : b, ee = e > S.W ? S.W : e;
                if (ee > bb) {
                    const int64_t cnt = ee - bb;
                    const int mpf = P.median_pre_filter;
                    FiltCache fc;
                    fc.c0 = -1;
                    auto fv = [&](int64_t q) -> float { return filtered_seq(S, mpf, bb + q, fc, L, gl); };
                    const float mean = np_sum_f32(fv, cnt, L, gl) / (float)cnt;
                    fc.c0 = -1;
                    auto dv = [&](int64_t q) -> float { const float x = filtered_seq(S, mpf, bb + q, fc, L, gl) - mean; return x * x; };
                    const float ss = np_sum_f32(dv, cnt, L, gl);
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
                const float dwell = np_sum_f32(pl, np_, L, gl);
                // spikes = the events of the interval that are not poly(A): np_ of its events are
                const int ns = (pj - pi + 1) - np_;
                pxg_polya_spike* rows = nullptr;           // lane 0 of the group owns the stores
                if (ns > 0 && sink.arena && gl == 0) {
                    const int base = atomicAdd(sink.cursor, ns);
                    out.spike_base = base;
                    if ((int64_t)base + ns <= (int64_t)sink.cap) rows = sink.arena + base;
                    else out.spill = 1;
                }
                int k = 0;
                for (int q = pi; q <= pj && k < ns; q++) {
                    if (!is_polya(q)) {
                        // the event fetches are group-uniform (all 16 lanes), the store is lane 0's
                        const float s0 = ev_at(q).length;
                        const float s1 = q - 1 >= pi ? ev_at(q - 1).mean : __builtin_nanf("");
                        const float s2 = ev_at(q).mean;
                        const float s3 = q + 1 <= pj ? ev_at(q + 1).mean : __builtin_nanf("");
                        if (rows) {
                            rows[k].v[0] = s0; rows[k].v[1] = s1;
                            rows[k].v[2] = s2; rows[k].v[3] = s3;
                        }
                        k++;
                    }
                }
                out.called = 1;
                out.begin = (int64_t)e0.start + ib;
                out.end = (int64_t)((double)e1.start + (double)e1.length) + ib;
                out.dwell = (int)dwell;
                out.n_spikes = ns;
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
            const float asum = np_sum_f32(aml, m, L, gl);
            ac = { -1, -1 };
            auto aln = [&](int64_t idx) -> float {
                return ev_at(nth_matching(anchor, 0, ne - 1, (int)idx, ac)).length;
            };
            const float pm = asum / np_sum_f32(aln, m, L, gl);
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
            const float tot = np_sum_f32(pl, np_, L, gl);
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
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PA_WAVES_PER_EU, 8))) void k_polya(int64_t n_reads, PolyaParams P,
                                              const int16_t* __restrict__ raw,
                                              const int64_t* __restrict__ off,
                                              const pxg_calib* __restrict__ cal,
                                              const float* __restrict__ ss,
                                              const int32_t* __restrict__ status,
                                              const int32_t* __restrict__ segs, Ev* __restrict__ evbuf,
                                              double2* __restrict__ snapbuf,
                                              int32_t* __restrict__ pout /* n x 8 */,
                                              pxg_polya_spike* __restrict__ spikes, int spike_cap,
                                              const int32_t* __restrict__ subset /* read ids of a retry pass, or null */,
                                              int32_t* __restrict__ over /* [0] count, [1] max events, [2] spike rows handed out, [PA_OVER_HEAD..] read ids */)
{
    __shared__ GroupLds lds[PXG_PA_LANES];
    const int grp = threadIdx.x / PA_GL, gl = threadIdx.x % PA_GL;     // read slot, lane inside it
    const int64_t slot = blockIdx.x * (int64_t)PXG_PA_LANES + grp;
    if (slot >= n_reads) return;
    const int64_t r = subset ? subset[slot] : slot;
    int32_t* po = pout + r * 8;
    if (gl == 0)
        for (int q = 0; q < 8; q++) po[q] = 0;
    const SpikeSink sink = { spikes, spike_cap, over + 2 };
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
    polya_one_read(P, raw + off[r], off[r + 1] - off[r], c.range / c.digitisation, c.offset,
                   ss[2 * r], ss[2 * r + 1], rb, re, has_end, &lds[grp], gl, grp, ev,
                   snapbuf + ((size_t)blockIdx.x * PXG_PA_LANES + grp) * PA_PRE, out, sink);
    if (gl != 0) return;
    if (out.overflow || out.spill) {
        over[PA_OVER_HEAD + atomicAdd(over, 1)] = (int32_t)r;
        atomicMax(over + 1, out.overflow);
    }
    po[0] = out.called;
    po[1] = out.n_spikes;
    po[7] = out.spike_base;
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
    __shared__ GroupLds lds[PXG_PA_LANES];
    const int grp = threadIdx.x / PA_GL, gl = threadIdx.x % PA_GL;
    const int64_t r = blockIdx.x * (int64_t)PXG_PA_LANES + grp;
    if (r >= n_windows) return;
    WindowSrc S;
    S.raw = nullptr; S.sig = sig + off[r]; S.ib = 0; S.W = off[r + 1] - off[r];
    S.k = 0; S.offset = 0; S.scale = 1; S.shift = 0;
    if (S.W <= 0) { if (gl == 0) n_events[r] = 0; return; }
    Ev* ev = evbuf + (size_t)blockIdx.x * P.ev_cap * PXG_PA_LANES;
    const int ne = detect_events_group(S, P, &lds[grp], gl, grp, ev, P.ev_cap, nullptr, nullptr);
    if (gl == 0) n_events[r] = ne;
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
        pxg_set_err(ctx, "poly(A) kernel supports median_pre_filter in {<=1, 7} and event windows <= 31");
        return PXG_E_UNSUPPORTED;
    }
    return PXG_OK;
}

#define PA_EV_CAP 4096            // event rows per read of the first pass
#define PA_SPIKES_PER_READ 2       // spike rows of the first pass, per read of the batch (the arena grows on demand)
#define PA_RETRY_BYTES (1ll << 31) // event scratch of one retry launch

// ---------------------------------------------------------------------------
// First pass in the order of the work: reads sorted by the length of their segmented poly(A) state, longest first
// (1 024 classes of two 15-sample blocks; a counting sort on the device).  The four reads of a wave run in lockstep --
// a wave takes as long as its longest window -- and all waves of a 10 000-read batch are resident at once, three to
// a SIMD: in batch order a wave's four windows are 1.4 k - 3.4 k samples at random and some SIMD holds three long
// waves; in this order a wave's windows are alike and a SIMD's three waves (blocks b, b + 1 024, b + 2 048 of the
// launch) are a long, a middling and a short one.  k_polya takes the order as the `subset` it already knows from
// the retry pass; every record is a function of its read alone.  PXG_POLYA_INPUT_ORDER=1: batch order (round 4).
#define PA_ORD_CLASSES 1024
__device__ __forceinline__ int polya_work_class(const PolyaParams& P, const int32_t* status, const int32_t* segs, int64_t r)
{
    if (status[r] != PXG_ST_OKAY) return PA_ORD_CLASSES - 1;                     // nothing to do: last
    const int32_t* first = segs + r * 2 * PXG_N_SEGMENTS;
    const int32_t* last = first + PXG_N_SEGMENTS;
    if (P.adapter_state < 0 || first[P.adapter_state] < 0) return PA_ORD_CLASSES - 1;
    if (P.polya_state < 0 || first[P.polya_state] < 0) return 0;                  // open end: windows grow by retries, first
    const int len = last[P.polya_state] - first[P.polya_state] + 1;
    const int c = len / 2;
    return PA_ORD_CLASSES - 2 - (c < 0 ? 0 : (c > PA_ORD_CLASSES - 3 ? PA_ORD_CLASSES - 3 : c));
}

__global__ __launch_bounds__(1024) void k_polya_order_count(int64_t n, PolyaParams P, const int32_t* __restrict__ status,
                                                            const int32_t* __restrict__ segs, int32_t* __restrict__ hist)
{
    __shared__ int h[PA_ORD_CLASSES];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r < n) atomicAdd(&h[polya_work_class(P, status, segs, r)], 1);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(1024) void k_polya_order_starts(int32_t* __restrict__ hist)
{
    __shared__ int a[2][PA_ORD_CLASSES];
    const int t = threadIdx.x;
    const int own = hist[t];
    a[0][t] = own;
    __syncthreads();
    int cur = 0;
    for (int d = 1; d < PA_ORD_CLASSES; d <<= 1) {
        a[cur ^ 1][t] = a[cur][t] + (t >= d ? a[cur][t - d] : 0);
        cur ^= 1;
        __syncthreads();
    }
    hist[t] = a[cur][t] - own;
}

__global__ __launch_bounds__(1024) void k_polya_order_place(int64_t n, PolyaParams P, const int32_t* __restrict__ status,
                                                            const int32_t* __restrict__ segs, int32_t* __restrict__ cursor,
                                                            int32_t* __restrict__ order)
{
    __shared__ int h[PA_ORD_CLASSES], base[PA_ORD_CLASSES];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int cls = 0, rank = 0;
    if (r < n) {
        cls = polya_work_class(P, status, segs, r);
        rank = atomicAdd(&h[cls], 1);
    }
    __syncthreads();
    if (h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (r < n) order[base[cls] + rank] = (int32_t)r;
}

static void launch_polya(pxg_ctx* ctx, int64_t n, int cap, const int32_t* subset, const int16_t* raw,
                         const int64_t* off, const pxg_calib* cal, const float* ss, const int32_t* status,
                         const int32_t* segs, int32_t* pout, DevBuf<pxg_polya_spike>& spikes)
{
    const int64_t blocks = (n + PXG_PA_LANES - 1) / PXG_PA_LANES;
    const size_t ev_bytes = (size_t)blocks * cap * PXG_PA_LANES * sizeof(Ev);
    const PolyaParams P = make_params(ctx->cfg, cap, ctx->cfg.polya_median_pre_filter);
    const int spike_cap = (int)std::min<size_t>(spikes.cap, 0x7fffffffu);
    hipLaunchKernelGGL(k_polya, dim3((unsigned)blocks), dim3(64), 0, ctx->stream, n, P, raw, off, cal,
                       ss, status, segs, (Ev*)ctx->polya_ev.p, (double2*)(ctx->polya_ev.p + ev_bytes),
                       pout, spikes.p, spike_cap, subset, ctx->polya_over.p);
}

static int reserve_polya(pxg_ctx* ctx, int64_t n, int cap)
{
    const int64_t blocks = (n + PXG_PA_LANES - 1) / PXG_PA_LANES;
    const size_t ev_bytes = (size_t)blocks * cap * PXG_PA_LANES * sizeof(Ev);
    const size_t snap_bytes = (size_t)blocks * PXG_PA_LANES * PA_PRE * sizeof(double2);
    return pxg_reserve(ctx, ctx->polya_ev, ev_bytes + snap_bytes);
}

// grow an arena and keep what it holds (the spike rows of the reads that fitted)
template <typename T>
static int reserve_keep(pxg_ctx* ctx, DevBuf<T>& b, size_t n, size_t used)
{
    if (n <= b.cap && b.p) return PXG_OK;
    DevBuf<T> bigger;
    int rc = pxg_reserve(ctx, bigger, n);
    if (rc) return rc;
    if (b.p && used) {
        hipError_t e = hipMemcpyAsync(bigger.p, b.p, std::min(used, b.cap) * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(bigger.p);
            pxg_set_err(ctx, std::string("spike arena copy: ") + hipGetErrorString(e));
            return PXG_E_HIP;
        }
    }
    if (b.p) (void)hipFree(b.p);
    b = bigger;
    return PXG_OK;
}

int pxg_launch_polya(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                     const pxg_calib* cal, const float* ss, const int32_t* status,
                     const int32_t* segs, int32_t* pout, DevBuf<pxg_polya_spike>& spikes)
{
    if (n <= 0) return PXG_OK;
    int rc = pxg_polya_supported(ctx);
    if (rc) return rc;
    if ((rc = reserve_polya(ctx, n, PA_EV_CAP))) return rc;
    if ((rc = pxg_reserve(ctx, ctx->polya_over, (size_t)n + PA_OVER_HEAD))) return rc;
    if ((rc = pxg_reserve(ctx, spikes, (size_t)n * PA_SPIKES_PER_READ + 1024))) return rc;
    PXG_HIP(ctx, hipMemsetAsync(ctx->polya_over.p, 0, PA_OVER_HEAD * sizeof(int32_t), ctx->stream));
    const int32_t* order = nullptr;
    if (!getenv("PXG_POLYA_INPUT_ORDER")) {             // (read per call: the tests take both orders)
        if ((rc = pxg_reserve(ctx, ctx->polya_order, (size_t)n + PA_ORD_CLASSES))) return rc;
        int32_t* hist = ctx->polya_order.p + n;
        PXG_HIP(ctx, hipMemsetAsync(hist, 0, PA_ORD_CLASSES * sizeof(int32_t), ctx->stream));
        const PolyaParams P = make_params(ctx->cfg, PA_EV_CAP, ctx->cfg.polya_median_pre_filter);
        const unsigned blocks = (unsigned)((n + 1023) / 1024);
        hipLaunchKernelGGL(k_polya_order_count, dim3(blocks), dim3(1024), 0, ctx->stream, n, P, status, segs, hist);
        hipLaunchKernelGGL(k_polya_order_starts, dim3(1), dim3(1024), 0, ctx->stream, hist);
        hipLaunchKernelGGL(k_polya_order_place, dim3(blocks), dim3(1024), 0, ctx->stream, n, P, status, segs, hist, ctx->polya_order.p);
        order = ctx->polya_order.p;
    }
    launch_polya(ctx, n, PA_EV_CAP, order, raw, off, cal, ss, status, segs, pout, spikes);
    return PXG_OK;
}

// The first pass gives every read PA_EV_CAP event rows and the batch PA_SPIKES_PER_READ spike
// rows per read.  A window that needs more event rows (tens of thousands of samples of
// open-ended extension over a featureless signal), or a tail whose spikes found the arena full,
// is listed by the kernel and re-run here with what it asked for: event rows doubled, so that a
// further extension still fits (the loop ends at the latest when the rows cover the longest
// read); the spike arena grown to twice the rows handed out so far, what it holds kept.
// Synchronises the stream; *retried = reads that went through a retry; *spike_rows = rows
// handed out (the arena's used prefix; the rows of re-run reads are simply left behind).
int pxg_polya_settle(pxg_ctx* ctx, int64_t n, int64_t longest_read, const int16_t* raw, const int64_t* off,
                     const pxg_calib* cal, const float* ss, const int32_t* status, const int32_t* segs,
                     int32_t* pout, DevBuf<pxg_polya_spike>& spikes, int64_t* retried, int64_t* spike_rows)
{
    if (retried) *retried = 0;
    if (spike_rows) *spike_rows = 0;
    if (n <= 0) return PXG_OK;
    int64_t cap = PA_EV_CAP;
    for (;;) {
        int32_t head[PA_OVER_HEAD] = { 0, 0, 0, 0 };
        PXG_HIP(ctx, hipMemcpyAsync(head, ctx->polya_over.p, sizeof(head), hipMemcpyDeviceToHost, ctx->stream));
        PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int64_t m = head[0];
        const int64_t handed_out = (int64_t)(uint32_t)head[2];
        if (spike_rows) *spike_rows = handed_out;
        if (m <= 0) return PXG_OK;
        if (retried) *retried += m;
        if (head[1] > 0) {                      // somebody ran out of event rows
            if (cap > longest_read) {
                pxg_set_err(ctx, "poly(A): event scratch overflow with rows for the longest read of the batch");
                return PXG_E_UNSUPPORTED;
            }
            cap = std::min<int64_t>(std::max<int64_t>(2 * (int64_t)head[1], 2 * cap), longest_read + 1);
        }
        if (handed_out >= (int64_t)0x3fffffff) {
            pxg_set_err(ctx, "poly(A): more than 2^30 spike rows in one batch");
            return PXG_E_UNSUPPORTED;
        }
        int rc;
        if (handed_out > (int64_t)spikes.cap &&
            (rc = reserve_keep(ctx, spikes, (size_t)(2 * handed_out + 1024), (size_t)handed_out)))
            return rc;
        // the listed reads move to a list of their own: the kernel appends the next round's to polya_over
        if ((rc = pxg_reserve(ctx, ctx->polya_retry, (size_t)m))) return rc;
        PXG_HIP(ctx, hipMemcpyAsync(ctx->polya_retry.p, ctx->polya_over.p + PA_OVER_HEAD, (size_t)m * sizeof(int32_t),
                                    hipMemcpyDeviceToDevice, ctx->stream));
        PXG_HIP(ctx, hipMemsetAsync(ctx->polya_over.p, 0, 2 * sizeof(int32_t), ctx->stream));   // (the row cursor runs on)
        const int64_t per_launch = std::max<int64_t>(
            PXG_PA_LANES, PA_RETRY_BYTES / (cap * (int64_t)sizeof(Ev)) / PXG_PA_LANES * PXG_PA_LANES);
        if ((rc = reserve_polya(ctx, std::min(per_launch, m), (int)cap))) return rc;
        for (int64_t at = 0; at < m; at += per_launch) {
            const int64_t k = std::min(per_launch, m - at);
            launch_polya(ctx, k, (int)cap, ctx->polya_retry.p + at, raw, off, cal, ss, status, segs, pout, spikes);
        }
    }
}

// Spike rows of a settled run in read order (CSR): offsets[r] .. offsets[r + 1] = rows of read r.
// offsets is always filled; rows only if cap_rows holds them all (PXG_E_NOMEM otherwise).
int pxg_polya_collect_spikes(pxg_ctx* ctx, int64_t n, const int32_t* pout, const DevBuf<pxg_polya_spike>& spikes,
                             int64_t spike_rows, int64_t cap_rows, pxg_polya_spike* out, int64_t* offsets)
{
    if (n <= 0) { if (offsets) offsets[0] = 0; return PXG_OK; }
    if (!offsets) { pxg_set_err(ctx, "spike offsets are required"); return PXG_E_INVALID; }
    std::vector<int32_t> po((size_t)n * 8);
    int rc = pxg_d2h_sync(ctx, po.data(), pout, po.size() * sizeof(int32_t));
    if (rc) return rc;
    offsets[0] = 0;
    for (int64_t r = 0; r < n; r++)
        offsets[r + 1] = offsets[r] + (po[(size_t)r * 8] ? (int64_t)po[(size_t)r * 8 + 1] : 0);
    const int64_t total = offsets[n];
    if (total > cap_rows || (total && !out)) {
        pxg_set_err(ctx, "spike rows: the buffer is smaller than the sum of the records' polya_n_spikes");
        return PXG_E_NOMEM;
    }
    if (!total) return PXG_OK;
    const int64_t used = std::min<int64_t>(spike_rows, (int64_t)spikes.cap);
    std::vector<pxg_polya_spike> arena((size_t)used);
    if ((rc = pxg_d2h_sync(ctx, arena.data(), spikes.p, (size_t)used * sizeof(pxg_polya_spike)))) return rc;
    for (int64_t r = 0; r < n; r++) {
        const int64_t ns = offsets[r + 1] - offsets[r], base = po[(size_t)r * 8 + 7];
        if (!ns) continue;
        if (base < 0 || base + ns > used) {
            pxg_set_err(ctx, "spike rows: a read's rows lie outside the arena (unsettled run?)");
            return PXG_E_STATE;
        }
        memcpy(out + offsets[r], arena.data() + base, (size_t)ns * sizeof(pxg_polya_spike));
    }
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

