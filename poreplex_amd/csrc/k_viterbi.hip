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
// No back-pointer table: the segmentation model is left-to-right (every edge
// i->j has j >= i), so a path is fully described by the step at which it
// entered each state.  Each lane carries those entry steps packed 16 bit per
// state; taking an in-edge copies the source lane's vector and stamps the
// current step.  The run-length summary falls out without a traceback.
#include "pxg_common.h"

#define VIT_CHAINS 1            // independent 8-read sets per recurrence wave (2 measured slower: 2.6 vs 1.9 ms, the wave is issue-bound)
#define VIT_READS (8 * VIT_CHAINS)
#define VIT_CHUNK 16
#define VIT_THREADS (64 * (1 + 2 * VIT_CHAINS))   // wave 0: recurrence; the others: emissions of the next chunk

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

// stamp a 16-bit field (state q) of the packed entry vector without dynamic
// register indexing
__device__ __forceinline__ void ent_stamp(unsigned (&e)[4], int q, unsigned val16)
{
    const unsigned sh = (unsigned)(q & 1) * 16u;
    const int w = q >> 1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned nv = (e[i] & ~(0xFFFFu << sh)) | (val16 << sh);
        e[i] = (i == w) ? nv : e[i];
    }
}

#define EM_STRIDE (VIT_CHUNK * PXG_MAX_STATES + 8)   // +8 doubles: spread reads over banks

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

// RAW=true : signal is pooled on the fly from int16 DAQ samples
// RAW=false: signal is an already pooled+scaled float arena (test hook)
// SPANS: bit k set = some edge goes from state s-k to state s (k >= 1);
// NW: packed entry words in use = ceil(n_states / 2)
template <bool RAW, unsigned SPANS, int NW>
__global__ __launch_bounds__(VIT_THREADS) void k_viterbi_ltr(
    int64_t n_reads, PxgHmmDev H, const int16_t* __restrict__ raw, const float* __restrict__ sig,
    const int64_t* __restrict__ off, const pxg_calib* __restrict__ cal,
    const float* __restrict__ ss, int stride, int scan_pooled, int32_t* __restrict__ status,
    int32_t* __restrict__ segs, double* __restrict__ logp_out, const double* __restrict__ lsetab_g)
{
    __shared__ double em[2][VIT_READS * EM_STRIDE];   // double buffer: producers run one chunk ahead
    __shared__ double lsetab[PXG_LSE_TAB_DOUBLES];
    for (int i = threadIdx.x; i < PXG_LSE_TAB_DOUBLES; i += blockDim.x) lsetab[i] = lsetab_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int rr = lane >> 3, s = lane & 7;
    const int S = H.n_states;

    // the recurrence wave carries VIT_CHAINS independent sets of 8 reads (chain c = reads
    // [8c, 8c + 8) of the block); two interleaved chains were measured slower than one
    int64_t r[VIT_CHAINS];
    bool valid_read[VIT_CHAINS];
    int T[VIT_CHAINS];
    int Tmax = 0;
#pragma unroll
    for (int c = 0; c < VIT_CHAINS; c++) {
        r[c] = blockIdx.x * (int64_t)VIT_READS + c * 8 + rr;
        valid_read[c] = r[c] < n_reads && (status == nullptr || status[r[c]] == PXG_ST_OKAY);
        T[c] = 0;
        if (valid_read[c]) {
            const int64_t len = off[r[c] + 1] - off[r[c]];
            const int64_t P = RAW ? len / stride : len;
            T[c] = (int)(P < scan_pooled ? P : scan_pooled);
        }
        Tmax = T[c] > Tmax ? T[c] : Tmax;
    }
    // longest read of this block (every wave computes the same value)
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(Tmax, d);
        Tmax = o > Tmax ? o : Tmax;
    }
    const int n_chunks = (Tmax + VIT_CHUNK - 1) / VIT_CHUNK;

    if (wv > 0) {
        // ================= emission producers (waves 1 .. 2*VIT_CHAINS) ===============
        // lane -> (read, step) of a chunk: 64 lanes = 4 reads x 16 steps;
        // consecutive lanes pool consecutive 15-sample blocks (coalesced)
        const int item = (wv - 1) * 64 + lane;
        const int prr = item / VIT_CHUNK, ptt = item % VIT_CHUNK;
        const int64_t pr = blockIdx.x * (int64_t)VIT_READS + prr;
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
        for (int c = 0; c <= n_chunks; c++) {
            if (c < n_chunks) {
                const int t = c * VIT_CHUNK + ptt;
                if (t < pT) {
                    float x;
                    if (RAW) {
                        float m = pxg_block_mean(raw + base + (int64_t)t * stride, stride, k, offset);
                        float y = scale * m;
                        x = y + shift;
                    } else {
                        x = sig[base + t];
                    }
                    const double xd = (double)x;
                    double* dst = &em[c & 1][prr * EM_STRIDE + ptt * PXG_MAX_STATES];
#pragma unroll
                    for (int q = 0; q < PXG_MAX_STATES; q++)
                        if (q < S) dst[q] = hmm_emission(H, lsetab, q, xd);
                }
            }
            __syncthreads();        // chunk c is published; chunk c-1 has been consumed
        }
        return;
    }

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
    const double lstart = (s < S) ? H.log_start[s] : -__builtin_inf();
    // lane-constant masks to stamp the 16-bit entry field of state s
    unsigned keep[4], put[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const unsigned field = 0xFFFFu << ((unsigned)(s & 1) * 16u);
        put[w] = (w == (s >> 1)) ? field : 0u;
        keep[w] = ~put[w];
    }
    const unsigned stamp_sh = (unsigned)(s & 1) * 16u;

    double v[VIT_CHAINS];
    unsigned ent[VIT_CHAINS][4];            // 16-bit entry step + 1 per state
#pragma unroll
    for (int c = 0; c < VIT_CHAINS; c++) {
        v[c] = -__builtin_inf();
#pragma unroll
        for (int w = 0; w < 4; w++) ent[c][w] = 0u;
    }

    __syncthreads();                // chunk 0 is in em[0]
    for (int c0 = 0; c0 < Tmax; c0 += VIT_CHUNK) {
        const double* emc = em[(c0 / VIT_CHUNK) & 1];
        // ---- recurrence phase ---------------------------------------------
        const int tend = (Tmax - c0) < VIT_CHUNK ? (Tmax - c0) : VIT_CHUNK;
#pragma unroll 1
        for (int tt = 0; tt < tend; tt++) {
            const int t = c0 + tt;
#pragma unroll
            for (int c = 0; c < VIT_CHAINS; c++) {
                const bool act = (t < T[c]) && (s < S);
                const double e = act ? emc[(c * 8 + rr) * EM_STRIDE + tt * PXG_MAX_STATES + s] : 0.0;
                // v and the entry vectors of the lanes below, for the spans in use
                double vs[PXG_MAX_STATES];
                unsigned es[PXG_MAX_STATES][NW];
#pragma unroll
                for (int k = 1; k < PXG_MAX_STATES; k++) {
                    if ((SPANS >> k) & 1u) {
                        vs[k] = dpp_shr_f64(v[c], k);
#pragma unroll
                        for (int w = 0; w < NW; w++) es[k][w] = dpp_shr_u32(ent[c][w], k);
                    }
                }
                if (t == 0) {                       // wave-uniform
                    if (act) {
                        v[c] = lstart + e;
#pragma unroll
                        for (int w = 0; w < NW; w++)
                            ent[c][w] = pxg_bfi(put[w], 1u << stamp_sh, ent[c][w]);
                    }
                } else {
                    double best = v[c] + lpk[0];    // span 0 = self loop (or -inf)
                    int bd = 0, bpr = prk[0];
#pragma unroll
                    for (int k = 1; k < PXG_MAX_STATES; k++) {
                        if ((SPANS >> k) & 1u) {
                            const double cand = vs[k] + lpk[k];
                            // three plain compares into SGPR masks (the short-circuit form
                            // compiled into nested exec-mask branches)
                            const unsigned long long take =
                                __ballot(cand > best) | (__ballot(cand == best) & __ballot(prk[k] < bpr));
                            best = pxg_sel_f64(take, best, cand);
                            bd = (int)pxg_sel_u32(take, (unsigned)bd, (unsigned)k);
                            bpr = (int)pxg_sel_u32(take, (unsigned)bpr, (unsigned)prk[k]);
                        }
                    }
                    const unsigned long long mact = __ballot(act);
                    v[c] = pxg_sel_f64(mact, v[c], best + e);
                    unsigned ne[NW];
#pragma unroll
                    for (int w = 0; w < NW; w++) ne[w] = ent[c][w];
#pragma unroll
                    for (int k = 1; k < PXG_MAX_STATES; k++) {
                        if ((SPANS >> k) & 1u) {
                            const unsigned long long mk = __ballot(bd == k);
#pragma unroll
                            for (int w = 0; w < NW; w++) ne[w] = pxg_sel_u32(mk, ne[w], es[k][w]);
                        }
                    }
                    const unsigned long long mmove = __ballot(act && bd != 0);
                    const unsigned stamp = (unsigned)(t + 1) << stamp_sh;
#pragma unroll
                    for (int w = 0; w < NW; w++)
                        ent[c][w] = pxg_sel_u32(mmove, ent[c][w], pxg_bfi(put[w], stamp, ne[w]));
                }
            }
        }
        __syncthreads();            // this chunk is consumed; the next one is published
    }

#pragma unroll
    for (int c = 0; c < VIT_CHAINS; c++) {
        // ---- termination: first maximum of the last column in name-sorted order -
        double bestv = -__builtin_inf();
        int end_lane = rr * 8 + H.order[0];
        for (int q = 0; q < S; q++) {
            const int ln = rr * 8 + H.order[q];
            const double vk = shfl_f64(v[c], ln);
            if (q == 0 || vk > bestv) {
                bestv = vk;
                end_lane = ln;
            }
        }
        unsigned fe[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
        for (int w = 0; w < NW; w++) fe[w] = (unsigned)__shfl((int)ent[c][w], end_lane);

        if (s == 0 && r[c] < n_reads) {
            int32_t* first = segs + r[c] * 2 * PXG_N_SEGMENTS;
            int32_t* last = first + PXG_N_SEGMENTS;
            for (int q = 0; q < PXG_N_SEGMENTS; q++) first[q] = last[q] = -1;
            if (valid_read[c] && T[c] > 0) {
                int prev = -1;
#pragma unroll
                for (int q = 0; q < PXG_MAX_STATES; q++) {
                    const int en = (int)((fe[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu);
                    if (q >= S || en == 0) continue;
                    first[q] = en - 1;
                    if (prev >= 0) last[prev] = en - 2;
                    prev = q;
                }
                if (prev >= 0) last[prev] = T[c] - 1;
                if (logp_out) logp_out[r[c]] = bestv;
                if (status != nullptr && H.adapter_state >= 0 && first[H.adapter_state] < 0)
                    status[r[c]] = PXG_ST_ADAPTER_NOT_DETECTED;
            } else if (logp_out) {
                logp_out[r[c]] = -__builtin_inf();
            }
        }
    }
}

static int check_supported(pxg_ctx* ctx, int which)
{
    const PxgHmmDev& H = ctx->hmm[which];
    if (!H.left_to_right) {
        ctx->err = "Viterbi kernel: HMM is not left-to-right (models with back-edges run through "
                   "the back-pointer scan of pxg_batch_unsplit_scan)";
        return PXG_E_UNSUPPORTED;
    }
    if (ctx->cfg.segmentation_scan_limit / ctx->cfg.stride >= 65535) {
        ctx->err = "segmentation_scan_limit/stride must be < 65535";
        return PXG_E_UNSUPPORTED;
    }
    return PXG_OK;
}

int pxg_launch_segment_raw(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                           const pxg_calib* cal, const float* ss, const int32_t* status,
                           int32_t* segs)
{
    if (n <= 0) return PXG_OK;
    int rc = check_supported(ctx, 0);
    if (rc) return rc;
    const int scan = ctx->cfg.segmentation_scan_limit / ctx->cfg.stride;
    const dim3 grid((unsigned)((n + VIT_READS - 1) / VIT_READS));
    const PxgHmmDev& H = ctx->hmm[0];
    if ((H.shift_mask & ~7u) == 0 && H.n_states <= 6)      // spans {1,2}: the shipped model
        hipLaunchKernelGGL((k_viterbi_ltr<true, 0x6u, 3>), grid, dim3(VIT_THREADS), 0, ctx->stream, n, H, raw,
                           (const float*)nullptr, off, cal, ss, ctx->cfg.stride, scan,
                           (int32_t*)status, segs, (double*)nullptr, ctx->d_lsetab);
    else
        hipLaunchKernelGGL((k_viterbi_ltr<true, 0xFEu, 4>), grid, dim3(VIT_THREADS), 0, ctx->stream, n, H, raw,
                           (const float*)nullptr, off, cal, ss, ctx->cfg.stride, scan,
                           (int32_t*)status, segs, (double*)nullptr, ctx->d_lsetab);
    return PXG_OK;
}

int pxg_launch_viterbi_f32(pxg_ctx* ctx, int which, int64_t n, const float* sig,
                           const int64_t* off, int32_t* segs, double* logp)
{
    if (n <= 0) return PXG_OK;
    int rc = check_supported(ctx, which);
    if (rc) return rc;
    const dim3 grid((unsigned)((n + VIT_READS - 1) / VIT_READS));
    const PxgHmmDev& H = ctx->hmm[which];
    if ((H.shift_mask & ~7u) == 0 && H.n_states <= 6)
        hipLaunchKernelGGL((k_viterbi_ltr<false, 0x6u, 3>), grid, dim3(VIT_THREADS), 0, ctx->stream, n, H,
                           (const int16_t*)nullptr, sig, off, (const pxg_calib*)nullptr,
                           (const float*)nullptr, 1, 65534, (int32_t*)nullptr, segs, logp, ctx->d_lsetab);
    else
        hipLaunchKernelGGL((k_viterbi_ltr<false, 0xFEu, 4>), grid, dim3(VIT_THREADS), 0, ctx->stream, n, H,
                           (const int16_t*)nullptr, sig, off, (const pxg_calib*)nullptr,
                           (const float*)nullptr, 1, 65534, (int32_t*)nullptr, segs, logp, ctx->d_lsetab);
    return PXG_OK;
}
