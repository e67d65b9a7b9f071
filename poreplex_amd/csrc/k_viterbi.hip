// k_viterbi.hip -- K3: pool + scale + 6-state HMM Viterbi + run-length summary
// (a1, a5, a7, a8: signal_loader.py:233-264; pomegranate viterbi called at
// signal_analyzer.py:352; run summary signal_analyzer.py:354-362).
//
// Layout: one block = 8 reads = 3 waves.  Wave 0 is 8 reads x 8 state-lanes and
// runs the recurrence; waves 1-2 produce the emissions one 16-step chunk ahead
// into a double-buffered LDS tile (measured: pooling 1.0 ms + densities 1.1 ms
// were serialised with a 1.7 ms recurrence).  Per chunk
//   (1) emission phase, all 64 lanes: lane (read, sub) pools 15 int16 samples
//       (NumPy pairwise order), applies fl(fl(scale*x)+shift), evaluates the
//       float64 log-densities of every state and parks them in LDS;
//   (2) recurrence phase: lane (read, state) keeps v[state] in a register and
//       pulls its in-edge sources from the lanes below it with DPP row shifts
//       (no LDS round trip; the model is left-to-right so a source is always
//       0..7 lanes down), max-plus with strict '>' in pomegranate's name-sorted
//       source order.
// Back-pointers: the segmentation model is left-to-right (every edge i->j has j >= i), so a
// step's decision is the SPAN of the in-edge it took (0 = self loop).  Each recurrence lane
// shifts that span into a register, FB bits per step, and stores the register once per
// 16-step chunk (one coalesced 256-byte row per wave).  The block's three waves then walk the
// fields of the winning states backwards, a wave per read, 64 chunks per load: the entry
// step of every state on the path IS the run-length summary.  (Round 1 carried the entry
// steps through the recurrence instead -- 6 DPP moves, 9 selects and 3 bit-field inserts
// of its ~66 instructions per step.)
#include <algorithm>
#include <type_traits>
#include "pxg_common.h"

#define VIT_READS 8             // reads per block = 8-lane groups of the recurrence wave
#define VIT_CHUNK 16
#define VIT_THREADS 192         // wave 0: recurrence; waves 1-2: emissions of the next chunk

__device__ __forceinline__ double hmm_emission(const PxgHmmDev& H, const double* lsetab, int s, double x)
{
    // pomegranate Normal: lssp - (x-mu)^2 * tss ; mixture: pair_lse fold
    double lp;
    {
        const double d = x - H.mu[s][0];
        lp = H.lssp[s][0] - (d * d) * H.tss[s][0];
    }
    if (H.n_mix[s] > 1) {
        lp = lp + H.logw[s][0];   // pair_lse(-inf, y) = y
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

__device__ __forceinline__ double shfl_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src);
    hi = __shfl(hi, src);
    return __hiloint2double(hi, lo);
}

// Workgroup barrier that publishes LDS only.  __syncthreads() also drains the vector-memory
// counter (its fence covers global memory), which would wait for the producers' prefetch of
// the NEXT chunk's samples at every chunk; nothing in this kernel passes global data
// between waves.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Emission tile of one chunk: em[read][state][step], a state's 16 steps + 1 double of padding, a read's 8 states
// = 136 doubles.  Round 6: step-minor.  (It was [read][step][state]: the 64 producer lanes of one state's store sat
// 64 bytes apart -- four bank positions, a 16-way conflict on every ds_write_b64, 75 % of K3's LDS cycles -- and the
// recurrence wave's reads queued behind them.)  Now a producer wave is lane = 4 step + read: a half wave's stores
// cover 32 distinct bank pairs (2 step + 16 read), and the recurrence lane (read, state) reads consecutive doubles
// from its own row, bank pairs 16 read + 34 state (mod 64) -- all distinct within a half wave.
#define EM_ROW (VIT_CHUNK + 1)
#define EM_STRIDE (PXG_MAX_STATES * EM_ROW)

// value of the lane `k` below inside a 16-lane row (DPP row_shr:k).  bound_ctrl: a lane
// whose source falls outside the row reads 0 -- such a lane has no edge of that span
// (lpk = -inf), so the value is never used, and the instruction needs no copy of an
// "old" operand (that copy was 10 of the ~66 instructions of a recurrence step).
template <typename T>
__device__ __forceinline__ int dpp_shr_i32(int v, T k)
{
    switch (k) {
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    case 3: return __builtin_amdgcn_update_dpp(0, v, 0x113, 0xF, 0xF, true);
    case 4: return __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    case 5: return __builtin_amdgcn_update_dpp(0, v, 0x115, 0xF, 0xF, true);
    case 6: return __builtin_amdgcn_update_dpp(0, v, 0x116, 0xF, 0xF, true);
    default: return __builtin_amdgcn_update_dpp(0, v, 0x117, 0xF, 0xF, true);
    }
}
__device__ __forceinline__ unsigned dpp_shr_u32(unsigned v, int k) { return (unsigned)dpp_shr_i32((int)v, k); }
__device__ __forceinline__ double dpp_shr_f64(double v, int k)
{
    const int lo = dpp_shr_i32(__double2loint(v), k), hi = dpp_shr_i32(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

// BT: register of back-pointer fields of one chunk -- uint32_t = 2 bits per step (spans <= 3),
//     uint64_t = 4 bits per step (spans <= 7)
template <typename BT> struct BpFields { static constexpr int bits = (int)sizeof(BT) * 8 / VIT_CHUNK; };

// A whole wave walks one read's back-pointer fields from the winning state at T-1 down to the
// start.  For the state the path is in, 64 lanes fetch its fields of 64 consecutive chunks
// (newest first); the newest non-zero field at or before the current step is the step the
// state was entered at and the span it was entered over.
template <typename BT>
__device__ __forceinline__ void viterbi_trace(const PxgHmmDev& H, const BT* __restrict__ mine /* this read's lane 0 */,
                                              int T, int cur, int lane, int32_t* __restrict__ first,
                                              int32_t* __restrict__ status_of_read)
{
    constexpr int FB = BpFields<BT>::bits;
    int32_t* last = first + PXG_N_SEGMENTS;
    if (cur < 0) {                                // not run, or no samples
        if (lane < 2 * PXG_N_SEGMENTS) first[lane] = -1;
        return;
    }
    int f[PXG_MAX_STATES], l[PXG_MAX_STATES];
#pragma unroll
    for (int q = 0; q < PXG_MAX_STATES; q++) f[q] = l[q] = -1;
    auto put = [&](int (&a)[PXG_MAX_STATES], int q, int val) {
#pragma unroll
        for (int i = 0; i < PXG_MAX_STATES; i++) a[i] = (i == q) ? val : a[i];
    };
    put(l, cur, T - 1);
    int t = T - 1;
    for (int hops = 0; hops < PXG_MAX_STATES; hops++) {
        int entered = 0, span = 0;                // entered stays 0: the path started in `cur`
        bool found = false;
        for (int cb = t / VIT_CHUNK; cb >= 0 && !found; cb -= 64) {
            const int c = cb - lane;
            BT w = c >= 0 ? mine[(size_t)c * 64 + cur] : (BT)0;
            if (c == t / VIT_CHUNK)               // drop the steps after t
                w &= ~(BT)0 << ((VIT_CHUNK - 1 - t % VIT_CHUNK) * FB);
            const unsigned long long nz = __ballot(w != 0);
            if (nz) {
                const int src = __builtin_ctzll(nz);          // lowest lane = newest chunk
                const int low = (sizeof(BT) == 8 ? __builtin_ctzll((unsigned long long)w | (w == 0))
                                                 : __builtin_ctz((unsigned)w | (w == 0))) / FB;
                const int my_t = c * VIT_CHUNK + (VIT_CHUNK - 1 - low);
                const int my_span = (int)((w >> (low * FB)) & (BT)((1 << FB) - 1));
                entered = __shfl(my_t, src);
                span = __shfl(my_span, src);
                found = true;
            }
        }
        put(f, cur, entered);
        if (!found) break;
        cur -= span;
        put(l, cur, entered - 1);
        t = entered - 1;
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < PXG_MAX_STATES; q++)
            if (q < PXG_N_SEGMENTS) { first[q] = f[q]; last[q] = l[q]; }
        if (status_of_read != nullptr && H.adapter_state >= 0 && f[H.adapter_state] < 0)
            *status_of_read = PXG_ST_ADAPTER_NOT_DETECTED;
    }
}

// RAW=true : signal is pooled on the fly from int16 DAQ samples; `sig`, if not null, is K1's
//            left-padded head tensor [n][head_width] -- the block means of the first
//            min(n_raw, head_limit) samples, by the same function -- and is read instead of
//            pooling those samples a second time
// RAW=false: signal is an already pooled+scaled float arena (test hook)
// SPANS: bit k set = some edge goes from state s-k to state s (k >= 1)
// BT: see BpFields
// POOL: 15 = the pooling stride is 15 (samples are prefetched into registers a chunk ahead),
//       0 = any stride / float input
template <bool RAW, int POOL, unsigned SPANS, typename BT>
__global__ __launch_bounds__(VIT_THREADS) void k_viterbi_ltr(
    int64_t n_reads, PxgHmmDev H, const int16_t* __restrict__ raw, const float* __restrict__ sig,
    const int64_t* __restrict__ off, const pxg_calib* __restrict__ cal,
    const float* __restrict__ ss, int stride, int scan_pooled, int head_width, int head_limit,
    int32_t* __restrict__ status, BT* __restrict__ bp /* [block][chunk][64 lanes] */, int bp_chunks,
    int32_t* __restrict__ segs, double* __restrict__ logp_out, const double* __restrict__ lsetab_g,
    const int32_t* __restrict__ ord /* reads in the order blocks take them (longest first), or null */)
{
    constexpr int FB = BpFields<BT>::bits;
    static_assert((SPANS >> (1 << FB)) == 0, "span does not fit the back-pointer field");
    __shared__ double em[2][VIT_READS * EM_STRIDE];   // double buffer: producers run one chunk ahead
    __shared__ double lsetab[PXG_LSE_TAB_DOUBLES];
    __shared__ int end_of[VIT_READS], steps_of[VIT_READS];     // winning state (-1: not run) and T per read
    for (int i = threadIdx.x; i < PXG_LSE_TAB_DOUBLES; i += blockDim.x) lsetab[i] = lsetab_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int rr = lane >> 3, s = lane & 7;
    const int S = H.n_states;

    const int64_t slot = blockIdx.x * (int64_t)VIT_READS + rr;
    const int64_t r = slot < n_reads ? (ord ? ord[slot] : slot) : n_reads;
    const bool valid_read = r < n_reads && (status == nullptr || status[r] == PXG_ST_OKAY);
    int T = 0;
    if (valid_read) {
        const int64_t len = off[r + 1] - off[r];
        const int64_t P = RAW ? len / stride : len;
        T = (int)(P < scan_pooled ? P : scan_pooled);
    }
    int Tmax = T;
    // longest read of this block (every wave computes the same value)
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(Tmax, d);
        Tmax = o > Tmax ? o : Tmax;
    }
    Tmax = __builtin_amdgcn_readfirstlane(Tmax);
    const int n_chunks = (Tmax + VIT_CHUNK - 1) / VIT_CHUNK;

    if (wv > 0) {
        // ================= emission producers (waves 1 and 2) ==========================
        // lane -> (read, step) of a chunk: 64 lanes = 16 steps x 4 reads, read-minor (see EM_ROW);
        // the 16 lanes of a read pool consecutive 15-sample blocks
        const int prr = (wv - 1) * 4 + (lane & 3), ptt = lane >> 2;
        const int64_t pslot = blockIdx.x * (int64_t)VIT_READS + prr;
        const int64_t pr = pslot < n_reads ? (ord ? ord[pslot] : pslot) : n_reads;
        const bool pvalid = pr < n_reads && (status == nullptr || status[pr] == PXG_ST_OKAY);
        int pT = 0;
        int64_t base = 0;
        double k = 0.0, offset = 0.0;
        float scale = 1.0f, shift = 0.0f;
        if (pvalid) {
            base = off[pr];
            const int64_t len = off[pr + 1] - off[pr];
            const int64_t P = RAW ? len / stride : len;
            pT = (int)(P < scan_pooled ? P : scan_pooled);
            if (RAW) {
                const pxg_calib c = cal[pr];
                k = c.range / c.digitisation;
                offset = c.offset;
                scale = ss[2 * pr];
                shift = ss[2 * pr + 1];
            }
        }
        auto publish = [&](int c, float x) {
            const double xd = (double)x;
            double* dst = &em[c & 1][prr * EM_STRIDE + ptt];
#pragma unroll
            for (int q = 0; q < PXG_MAX_STATES; q++)
                if (q < S) dst[q * EM_ROW] = hmm_emission(H, lsetab, q, xd);
        };
        if (RAW && POOL == 15) {
            // the 15 samples of chunk c+1 are requested before chunk c is worked on: one
            // memory round trip per chunk was ~all of this wave's time (measured: producers
            // alone 1.06 ms of a 1.07 ms kernel, their instructions ~0.3 ms)
            int16_t nx[15];
            float nxh = 0.0f;
#pragma unroll
            for (int j = 0; j < 15; j++) nx[j] = 0;
            int pNM = 0;                   // steps whose block mean K1 left in the head tensor
            const float* hrow = sig;
            if (pvalid && sig != nullptr) {
                const int64_t len = off[pr + 1] - off[pr];
                pNM = (int)((len < head_limit ? len : head_limit) / 15);
                hrow = sig + pr * (int64_t)head_width + (head_width - pNM);
            }
            auto fetch = [&](int c) {
                const int t = c * VIT_CHUNK + ptt;
                if (t < pNM) {
                    nxh = hrow[t];
                } else if (t < pT) {
                    const int16_t* src = raw + base + (int64_t)t * 15;
#pragma unroll
                    for (int j = 0; j < 15; j++) nx[j] = src[j];
                }
            };
            if (n_chunks > 0) fetch(0);
            for (int c = 0; c <= n_chunks; c++) {
                if (c < n_chunks) {
                    int16_t cur[15];
                    const float curh = nxh;
#pragma unroll
                    for (int j = 0; j < 15; j++) cur[j] = nx[j];
                    if (c + 1 < n_chunks) fetch(c + 1);
                    const int t = c * VIT_CHUNK + ptt;
                    if (t < pT) {
                        float m;
                        if (t < pNM) m = curh;
                        else m = pxg_block_mean15(cur, k, offset);
                        const float y = scale * m;
                        publish(c, y + shift);
                    }
                }
                lds_barrier();      // chunk c is published; chunk c-1 has been consumed
            }
        } else if (!RAW) {
            float nx = 0.0f;
            if (n_chunks > 0 && ptt < pT) nx = sig[base + ptt];
            for (int c = 0; c <= n_chunks; c++) {
                if (c < n_chunks) {
                    const float cur = nx;
                    const int tn = (c + 1) * VIT_CHUNK + ptt;
                    if (c + 1 < n_chunks && tn < pT) nx = sig[base + tn];
                    if (c * VIT_CHUNK + ptt < pT) publish(c, cur);
                }
                lds_barrier();
            }
        } else {
            for (int c = 0; c <= n_chunks; c++) {
                const int t = c * VIT_CHUNK + ptt;
                if (c < n_chunks && t < pT) {
                    const float m = pxg_block_mean(raw + base + (int64_t)t * stride, stride, k, offset);
                    const float y = scale * m;
                    publish(c, y + shift);
                }
                lds_barrier();
            }
        }
    } else {
    // ===================== recurrence (wave 0) ====================================
    // per-lane edge table by SPAN: the source of span k is the lane k below.
    // lpk[k] = log P(state s-k -> s) (-inf if no such edge); prk[k] = position of
    // that source in pomegranate's name-sorted in-edge order (first maximum wins)
    double lpk[PXG_MAX_STATES];
    int prk[PXG_MAX_STATES];
#pragma unroll
    for (int k = 0; k < PXG_MAX_STATES; k++) {
        lpk[k] = -__builtin_inf();
        prk[k] = 99;
    }
#pragma unroll
    for (int d = 0; d < PXG_MAX_STATES; d++) {
        const int sidx = (s < S) ? H.in_src[s][d] : -1;
        const double lp = (s < S) ? H.in_logp[s][d] : -__builtin_inf();
#pragma unroll
        for (int k = 0; k < PXG_MAX_STATES; k++) {
            const bool hit = sidx >= 0 && (s - sidx) == k;
            lpk[k] = hit ? lp : lpk[k];
            prk[k] = hit ? d : prk[k];
        }
    }
    // ties: candidate k replaces the running best (reached over span j) iff prk[k] < prk[j].
    // Those comparisons are lane constants: before[k][j] = lanes where span k outranks span j,
    // and which span holds the best is a partition of the wave kept in scalar masks, so the
    // tie rule costs no vector instruction inside the loop.
    unsigned long long before[PXG_MAX_STATES][PXG_MAX_STATES];
#pragma unroll
    for (int k = 1; k < PXG_MAX_STATES; k++)
#pragma unroll
        for (int j = 0; j < k; j++)
            before[k][j] = (((SPANS >> k) & 1u) && (j == 0 || ((SPANS >> j) & 1u))) ? __ballot(prk[k] < prk[j]) : 0ull;
    const double lstart = (s < S) ? H.log_start[s] : -__builtin_inf();
    const unsigned long long state_lanes = __ballot(s < S);

    double v = -__builtin_inf();    // runs on past T-1 (on stale emissions, never read back) ...
    double vfin = v;                // ... the column of step T-1 is kept here
    BT* bpw = bp + (size_t)blockIdx.x * bp_chunks * 64 + lane;
    // lanes of unused states read the emission of the last state: finite, and never used
    const int es = s < S ? s : S - 1;

    // One step.  EXACT = false is the speculative form: a candidate replaces the best iff it
    // is strictly greater, which is the whole rule unless two candidates of a live lane are
    // EQUAL (then pomegranate's in-edge order decides).  Its dependent chain is
    // add -> cmp -> select per span, all vector; the exact form adds two scalar mask
    // operations per span in the middle of that chain.  `ties` collects the live lanes that
    // saw an equality; a chunk that saw one is run again with EXACT = true (the first chunk,
    // while states are still unreachable at -inf, and practically never after it).
    // INNER (round 6): a chunk in which no lane's read ends -- every lane is live for all of its steps or dead for all of
    // them, so the live mask is the chunk's (`live_c`, a loop-invariant scalar pair) and nobody's last column has to be
    // kept: one v_cmp + one scalar AND (a VALU -> SALU round trip in an in-order wave) and a v_cmp + two v_cndmask
    // less per step, in all but <= 8 of a block's ~250 chunks.
    unsigned long long live_c = 0ull;
    auto step = [&](auto exact, auto inner, int t, double e, unsigned long long& ties, BT& fields) {
        constexpr bool EXACT = decltype(exact)::value, INNER = decltype(inner)::value;
        const unsigned long long live = INNER ? live_c : (__ballot(t < T) & state_lanes);
        double cand[PXG_MAX_STATES];
#pragma unroll
        for (int k = 1; k < PXG_MAX_STATES; k++)
            if ((SPANS >> k) & 1u) cand[k] = dpp_shr_f64(v, k) + lpk[k];
        double best = v + lpk[0];    // span 0 = self loop (or -inf)
        unsigned bd = 0u;
        if (EXACT) {
            unsigned long long holds[PXG_MAX_STATES];      // holds[j]: lanes whose best came over span j
            holds[0] = ~0ull;
#pragma unroll
            for (int k = 1; k < PXG_MAX_STATES; k++) {
                holds[k] = 0ull;
                if ((SPANS >> k) & 1u) {
                    unsigned long long outranks = 0ull;
#pragma unroll
                    for (int j = 0; j < k; j++) outranks |= holds[j] & before[k][j];
                    const unsigned long long take =
                        __ballot(cand[k] > best) | (__ballot(cand[k] == best) & outranks);
                    best = pxg_sel_f64(take, best, cand[k]);
                    bd = pxg_sel_u32(take, bd, (unsigned)k);
#pragma unroll
                    for (int j = 0; j < k; j++) holds[j] &= ~take;
                    holds[k] = take;
                }
            }
        } else {
#pragma unroll
            for (int k = 1; k < PXG_MAX_STATES; k++) {
                if ((SPANS >> k) & 1u) {
                    // (round 6) the running best is a v_max_f64 chain: add -> max -> max -> add per step; the
                    // comparisons that choose the span and flag ties hang off that chain instead of sitting in
                    // it (v_cmp -> SGPR pair -> two v_cndmask per span was 4 of its 7 dependent stages).  Without
                    // a tie max(best, cand) IS the strict-greater select; a chunk with a tie is replayed exactly.
                    ties |= INNER ? __ballot(cand[k] == best) : (__ballot(cand[k] == best) & live);      // (INNER: masked once, after the chunk)
                    bd = cand[k] > best ? (unsigned)k : bd;
#ifdef VIT_NO_MAX
                    best = cand[k] > best ? cand[k] : best;
#else
                    best = pxg_max_f64(best, cand[k]);
#endif
                }
            }
        }
        v = best + e;
        if (!INNER) vfin = pxg_sel_f64(__ballot(t == T - 1), vfin, v);
        fields = (fields << FB) | (BT)bd;          // steps at or after T leave fields nobody reads
    };
    auto run_chunk = [&](auto exact, auto inner, const double* emc, int c0, int tend, unsigned long long& ties) {
        BT fields = 0;
        int tt = 0;
        if (c0 == 0) {              // t = 0: out of the silent start state, no back-pointer
            v = (0 < T && s < S) ? lstart + emc[0] : v;
            vfin = pxg_sel_f64(__ballot(T == 1), vfin, v);
            tt = 1;
        }
#ifndef VIT_UNROLL
#define VIT_UNROLL 1
#endif
        const int n_steps = __builtin_amdgcn_readfirstlane(tend);      // (a scalar trip count: the compiler kept it in a vector register)
#pragma unroll VIT_UNROLL
        for (; tt < n_steps; tt++) step(exact, inner, c0 + tt, emc[tt], ties, fields);
        return (BT)(fields << ((VIT_CHUNK - tend) * FB));      // a short last chunk: step tt still sits at field 15 - tt
    };

    lds_barrier();                  // chunk 0 is in em[0]
    for (int c0 = 0; c0 < Tmax; c0 += VIT_CHUNK) {
        const double* emc = em[(c0 / VIT_CHUNK) & 1] + rr * EM_STRIDE + es * EM_ROW;
        const int tend = (Tmax - c0) < VIT_CHUNK ? (Tmax - c0) : VIT_CHUNK;
        const double v0 = v, vfin0 = vfin;
        unsigned long long ties = 0ull, unused = 0ull;
        live_c = __ballot(c0 < T) & state_lanes;
        const bool inner = c0 > 0 && __ballot(T - 1 >= c0 && T - 1 < c0 + tend) == 0ull;
        BT fields;
        if (inner) {
            fields = run_chunk(std::false_type(), std::true_type(), emc, c0, tend, ties);
            ties &= live_c;
        } else {
            fields = run_chunk(std::false_type(), std::false_type(), emc, c0, tend, ties);
        }
        if (ties != 0ull) {
            v = v0;
            vfin = vfin0;
            fields = run_chunk(std::true_type(), std::false_type(), emc, c0, tend, unused);
        }
        bpw[(size_t)(c0 / VIT_CHUNK) * 64] = fields;
        lds_barrier();              // this chunk is consumed; the next one is published
    }
    v = vfin;

    // ---- termination: first maximum of the last column in name-sorted order ------------
    double bestv = -__builtin_inf();
    int end_s = H.order[0];
    for (int q = 0; q < S; q++) {
        const double vk = shfl_f64(v, rr * 8 + H.order[q]);
        if (q == 0 || vk > bestv) {
            bestv = vk;
            end_s = H.order[q];
        }
    }
    if (s == 0) {
        const bool ran = valid_read && T > 0;
        end_of[rr] = ran ? end_s : -1;
        steps_of[rr] = T;
        if (logp_out && r < n_reads) logp_out[r] = ran ? bestv : -__builtin_inf();
    }
    }

    // ---- traceback: the three waves share the block's eight reads ---------------------------
    // (the fields were stored by wave 0 of this block and are read here for the first time in
    //  this kernel: __syncthreads() has waited for the stores, no line of them is in the L1)
    __syncthreads();
    for (int q = wv; q < VIT_READS; q += VIT_THREADS / 64) {
        const int64_t qslot = blockIdx.x * (int64_t)VIT_READS + q;
        if (qslot >= n_reads) break;
        const int64_t rq = ord ? ord[qslot] : qslot;
        viterbi_trace<BT>(H, bp + (size_t)blockIdx.x * bp_chunks * 64 + q * 8, steps_of[q], end_of[q], lane,
                          segs + rq * 2 * PXG_N_SEGMENTS, status ? status + rq : nullptr);
    }
}

static int check_supported(pxg_ctx* ctx, int which)
{
    const PxgHmmDev& H = ctx->hmm[which];
    if (!H.left_to_right) {
        pxg_set_err(ctx, "Viterbi kernel: HMM is not left-to-right (models with back-edges run through "
                   "the back-pointer scan of pxg_batch_unsplit_scan)");
        return PXG_E_UNSUPPORTED;
    }
    return PXG_OK;
}

// spans {1,2,3} fit 2-bit fields (the shipped model uses {1,2}); anything wider takes 4 bits
template <bool RAW>
static int launch_viterbi(pxg_ctx* ctx, const PxgHmmDev& H, int64_t n, const int16_t* raw, const float* sig,
                          const int64_t* off, const pxg_calib* cal, const float* ss, int stride, int scan,
                          int max_steps, int32_t* status, int32_t* segs, double* logp, const int32_t* ord = nullptr)
{
    const int head_width = ctx->cfg.scaler_length / ctx->cfg.stride, head_limit = ctx->cfg.scaler_length;
    const int64_t blocks = (n + VIT_READS - 1) / VIT_READS;
    const int bp_chunks = std::max(1, (std::min(scan, max_steps) + VIT_CHUNK - 1) / VIT_CHUNK);
    const bool narrow = (H.shift_mask & ~7u) == 0 && H.n_states <= 6;
    int rc = pxg_reserve(ctx, ctx->vit_bp, (size_t)blocks * bp_chunks * 64 * (narrow ? 4 : 8));
    if (rc) return rc;
    constexpr int GEN = 0;
    if (narrow) {
        uint32_t* bp = (uint32_t*)ctx->vit_bp.p;
        if (RAW && stride == 15)
            hipLaunchKernelGGL((k_viterbi_ltr<RAW, RAW ? 15 : GEN, 0x6u, uint32_t>), dim3((unsigned)blocks),
                               dim3(VIT_THREADS), 0, ctx->stream, n, H, raw, sig, off, cal, ss, stride, scan, head_width, head_limit, status,
                               bp, bp_chunks, segs, logp, ctx->d_lsetab, ord);
        else
            hipLaunchKernelGGL((k_viterbi_ltr<RAW, GEN, 0x6u, uint32_t>), dim3((unsigned)blocks),
                               dim3(VIT_THREADS), 0, ctx->stream, n, H, raw, sig, off, cal, ss, stride, scan, head_width, head_limit, status,
                               bp, bp_chunks, segs, logp, ctx->d_lsetab, ord);
    } else {
        uint64_t* bp = (uint64_t*)ctx->vit_bp.p;
        if (RAW && stride == 15)
            hipLaunchKernelGGL((k_viterbi_ltr<RAW, RAW ? 15 : GEN, 0xFEu, uint64_t>), dim3((unsigned)blocks),
                               dim3(VIT_THREADS), 0, ctx->stream, n, H, raw, sig, off, cal, ss, stride, scan, head_width, head_limit, status,
                               bp, bp_chunks, segs, logp, ctx->d_lsetab, ord);
        else
            hipLaunchKernelGGL((k_viterbi_ltr<RAW, GEN, 0xFEu, uint64_t>), dim3((unsigned)blocks),
                               dim3(VIT_THREADS), 0, ctx->stream, n, H, raw, sig, off, cal, ss, stride, scan, head_width, head_limit, status,
                               bp, bp_chunks, segs, logp, ctx->d_lsetab, ord);
    }
    return PXG_OK;
}

int pxg_launch_segment_raw(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                           const pxg_calib* cal, const float* ss, const float* head_or_null,
                           const int32_t* status, int32_t* segs, const int32_t* order)
{
    if (n <= 0) return PXG_OK;
    int rc = check_supported(ctx, 0);
    if (rc) return rc;
    const int scan = ctx->cfg.segmentation_scan_limit / ctx->cfg.stride;
    return launch_viterbi<true>(ctx, ctx->hmm[0], n, raw, head_or_null, off, cal, ss, ctx->cfg.stride, scan, scan,
                                (int32_t*)status, segs, nullptr, order);
}

int pxg_launch_viterbi_f32(pxg_ctx* ctx, int which, int64_t n, const float* sig,
                           const int64_t* off, int max_steps, int32_t* segs, double* logp)
{
    if (n <= 0) return PXG_OK;
    int rc = check_supported(ctx, which);
    if (rc) return rc;
    return launch_viterbi<false>(ctx, ctx->hmm[which], n, nullptr, sig, off, nullptr, nullptr, 1, 1 << 30,
                                 max_steps, nullptr, segs, logp);
}
