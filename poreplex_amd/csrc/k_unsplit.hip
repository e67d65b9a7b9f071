// k_unsplit.hip -- K7: Guppy event table (a18) and the window scan of the
// pseudo-fusion filter (a19).
//   a18  fast5_file.py:210-230  medfilt(5) of pA[first, first+15n) -> block means
//        signal_analyzer.py:318 scaled_mean = fl(fl(scale*mean)+shift)
//   a19  signal_analyzer.py:366-418: 8 s windows stepped 3 s over the event
//        table, Viterbi of the "unsplit" HMM (has back-edges), leader*/adapter run
//        analysis with duration cut-offs -> candidate in-read adapters.
//
// The unsplit HMM is not left-to-right, so this Viterbi keeps real back
// pointers: lane (read, state) packs its chosen source (4 bit/step) into a
// register and spills one dword per 8 steps to a wave-interleaved arena;
// the traceback walks it with one cross-lane read per step, the run analysis is
// a forward scan of the recovered path by one lane per read.
#include "pxg_common.h"

#define UN_READS 8
#define UN_CHUNK 32
#define UN_TMAX 4096                       // steps per window the scratch holds
#define UN_EM_STRIDE (UN_CHUNK * PXG_MAX_STATES + 8)

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

__global__ void k_guppy_event_means(int64_t n_reads, const int16_t* __restrict__ raw,
                                    const int64_t* __restrict__ off, const pxg_calib* __restrict__ cal,
                                    const float* __restrict__ ss, const int64_t* __restrict__ first_sample,
                                    const int64_t* __restrict__ ev_off, int stride,
                                    float* __restrict__ mean, float* __restrict__ scaled)
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
        float rsum[8];
        // pA of samples q-2 .. q+2 around each of the 15 block samples, zero outside [0, len)
        for (int j = 0; j < stride && j < 16; j++) {
            const int64_t q = e * stride + j;
            float v;
            if (q >= len) {
                v = __builtin_nanf("");
            } else {
                float w[5];
#pragma unroll
                for (int d = -2; d <= 2; d++) {
                    const int64_t p = q + d;
                    w[d + 2] = (p >= 0 && p < len) ? pxg_raw2pa(base[p], k, c.offset) : 0.0f;
                }
                v = med5(w[0], w[1], w[2], w[3], w[4]);
            }
            blk[j] = v;
        }
        // NumPy pairwise float32 sum of `stride` (= 15) values, then / stride
        float s;
        if (stride < 8) {
            s = 0.0f;
            for (int j = 0; j < stride; j++) s += blk[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) rsum[j] = blk[j];
            int i = 8;
            for (; i < stride - (stride % 8); i += 8)
                for (int j = 0; j < 8; j++) rsum[j] += blk[i + j];
            s = ((rsum[0] + rsum[1]) + (rsum[2] + rsum[3])) + ((rsum[4] + rsum[5]) + (rsum[6] + rsum[7]));
            for (; i < stride; i++) s += blk[i];
        }
        s = 0.0f + s;
        const float m = s / (float)stride;
        mean[ev_off[r] + e] = m;
        const float y = scale * m;
        scaled[ev_off[r] + e] = y + shift;
    }
}

int pxg_launch_guppy_event_means(pxg_ctx* ctx, int64_t n, const int16_t* raw, const int64_t* off,
                                 const pxg_calib* cal, const float* ss, const int64_t* first,
                                 const int64_t* ev_off, int stride, float* mean, float* scaled)
{
    if (n <= 0) return PXG_OK;
    if (stride < 1 || stride > 16) {
        ctx->err = "block_stride must be 1..16";
        return PXG_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_guppy_event_means, dim3((unsigned)n, 8), dim3(256), 0, ctx->stream, n, raw, off,
                       cal, ss, first, ev_off, stride, mean, scaled);
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

__device__ __forceinline__ double un_emission(const PxgHmmDev& H, int s, double x)
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
            const double a = lp, b = l;
            if (a == __builtin_inf() || b == __builtin_inf()) lp = __builtin_inf();
            else if (a == -__builtin_inf()) lp = b;
            else if (b == -__builtin_inf()) lp = a;
            else if (a > b) lp = a + log(exp(b - a) + 1.0);
            else lp = b + log(exp(a - b) + 1.0);
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

__global__ __launch_bounds__(64) void k_unsplit_scan(
    int64_t n_reads, PxgHmmDev H, UnsplitParams P, const pxg_calib* __restrict__ cal,
    const int32_t* __restrict__ status, const int32_t* __restrict__ segs,
    const int64_t* __restrict__ first_sample, const int64_t* __restrict__ ev_off,
    const float* __restrict__ scaled, unsigned* __restrict__ bpbuf /* [wave][UN_TMAX/8][64] */,
    unsigned char* __restrict__ pathbuf /* [wave][UN_TMAX][8] */,
    int64_t* __restrict__ out_iv /* n x PXG_MAX_UNSPLIT x 2 */, int32_t* __restrict__ out_cnt)
{
    __shared__ double em[UN_READS * UN_EM_STRIDE];
    const int lane = threadIdx.x;
    const int rr = lane >> 3, s = lane & 7;
    const int64_t r = blockIdx.x * (int64_t)UN_READS + rr;
    const int S = H.n_states;
    unsigned* bpw = bpbuf + (size_t)blockIdx.x * (UN_TMAX / 8) * 64;
    unsigned char* path = pathbuf + (size_t)blockIdx.x * UN_TMAX * 8;

    // ---- per-read window geometry ------------------------------------------
    bool valid = r < n_reads && status[r] == PXG_ST_OKAY;
    int64_t first = 0, n_ev = 0, payload_start = 0, last_end = 0;
    int64_t window_size = 0, window_step = 1, strict_duration = 0;
    int64_t cut_total[2] = { 0, 0 }, cut_adapter[2] = { 0, 0 };
    const float* x = scaled;
    if (valid) {
        const int32_t* sf = segs + r * 2 * PXG_N_SEGMENTS;
        const int a_last = sf[PXG_N_SEGMENTS + P.seg_adapter_state];
        n_ev = ev_off[r + 1] - ev_off[r];
        if (sf[P.seg_adapter_state] < 0 || n_ev <= 0) valid = false;
        const double rate = cal[r].sampling_rate;
        first = first_sample[r];
        payload_start = (int64_t)(a_last + 1) * P.pool_stride;          // :369
        last_end = first + (int64_t)P.stride * (n_ev - 1) + 1;
        window_size = (int64_t)(P.window_size * rate);
        window_step = (int64_t)(P.window_step * rate);
        strict_duration = (int64_t)(P.strict_duration * rate);
        cut_total[0] = (int64_t)(P.loosen_full * rate); cut_total[1] = (int64_t)(P.strict_full * rate);
        cut_adapter[0] = (int64_t)(P.loosen_dna * rate); cut_adapter[1] = (int64_t)(P.strict_dna * rate);
        if (window_step <= 0) valid = false;
        x = scaled + ev_off[r];
    }
    if (s == 0 && r < n_reads) out_cnt[r] = 0;

    // per-lane in-edge table (name-sorted slots)
    int src_lane[PXG_MAX_STATES], src_state[PXG_MAX_STATES];
    double src_lp[PXG_MAX_STATES];
#pragma unroll
    for (int d = 0; d < PXG_MAX_STATES; d++) {
        const int sidx = (s < S) ? H.in_src[s][d] : -1;
        src_state[d] = sidx >= 0 ? sidx : 7;
        src_lane[d] = sidx >= 0 ? (rr * 8 + sidx) : lane;
        src_lp[d] = sidx >= 0 ? H.in_logp[s][d] : -__builtin_inf();
    }
    const double lstart = (s < S) ? H.log_start[s] : -__builtin_inf();

    int count = 0;
    bool overflow = false;
    int64_t left = payload_start;
    bool more = valid && left < last_end;
    while (__any(more)) {
        // ---- this window's event range -------------------------------------
        int64_t k0 = 0, k1 = -1;
        if (more) {
            k0 = (left - first) <= 0 ? 0 : (left - first + P.stride - 1) / P.stride;
            k1 = (left + window_size - first) < 0 ? -1 : (left + window_size - first) / P.stride;
            if (k1 > n_ev - 1) k1 = n_ev - 1;
            if (k1 < k0) more = false;                      // empty block: stop this read (:387-388)
        }
        int T = more ? (int)(k1 - k0 + 1) : 0;
        if (T > UN_TMAX) { T = 0; more = false; overflow = true; }   // window longer than the scratch
        int Tmax = T;
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(Tmax, d);
            Tmax = o > Tmax ? o : Tmax;
        }
        if (Tmax == 0) break;

        // ---- forward pass ---------------------------------------------------
        double v = -__builtin_inf();
        unsigned bpacc = 0;
        for (int c0 = 0; c0 < Tmax; c0 += UN_CHUNK) {
            __syncthreads();
#pragma unroll 1
            for (int p = 0; p < UN_CHUNK / 8; p++) {
                const int tt = p * 8 + s;
                const int t = c0 + tt;
                if (t < T) {
                    const double xd = (double)x[k0 + t];
#pragma unroll
                    for (int q = 0; q < PXG_MAX_STATES; q++)
                        if (q < S) em[rr * UN_EM_STRIDE + tt * PXG_MAX_STATES + q] = un_emission(H, q, xd);
                }
            }
            __syncthreads();
            const int tend = (Tmax - c0) < UN_CHUNK ? (Tmax - c0) : UN_CHUNK;
#pragma unroll 1
            for (int tt = 0; tt < tend; tt++) {
                const int t = c0 + tt;
                const bool act = (t < T) && (s < S);
                const double e = act ? em[rr * UN_EM_STRIDE + tt * PXG_MAX_STATES + s] : 0.0;
                double best = -__builtin_inf();
                unsigned arg = 7u;
#pragma unroll
                for (int d = 0; d < PXG_MAX_STATES; d++) {
                    if (d < H.max_in) {                     // wave-uniform
                        const double vk = un_shfl_f64(v, src_lane[d]);
                        const double cand = vk + src_lp[d];
                        const unsigned long long take = __ballot(cand > best);
                        best = pxg_sel_f64(take, best, cand);
                        arg = pxg_sel_u32(take, arg, (unsigned)src_state[d]);
                    }
                }
                const unsigned long long mact = __ballot(act);
                const double nv = (t == 0) ? (lstart + e) : (best + e);
                v = pxg_sel_f64(mact, v, nv);
                bpacc |= (arg & 7u) << ((t & 7) * 4);
                if ((t & 7) == 7 || t == Tmax - 1) {
                    bpw[(t >> 3) * 64 + lane] = bpacc;
                    bpacc = 0;
                }
            }
        }
        // ---- termination + traceback ------------------------------------------
        double bestv = -__builtin_inf();
        int cur = H.order[0];
        for (int q = 0; q < S; q++) {
            const double vk = un_shfl_f64(v, rr * 8 + H.order[q]);
            if (q == 0 || vk > bestv) { bestv = vk; cur = H.order[q]; }
        }
        __syncthreads();      // bp words of this wave are visible to its own lanes
        for (int g = (Tmax - 1) >> 3; g >= 0; g--) {
            const unsigned w = bpw[g * 64 + lane];
            for (int q = 7; q >= 0; q--) {
                const int t = g * 8 + q;
                if (t >= Tmax) continue;
                const bool in = t < T;
                if (in && s == 0) path[(size_t)t * 8 + rr] = (unsigned char)cur;
                const int mine = (int)((w >> (q * 4)) & 7u);
                const int src = __shfl(mine, rr * 8 + cur);
                if (in && t > 0 && src != 7) cur = src;
            }
        }
        __syncthreads();
        // ---- run analysis (signal_analyzer.py:393-418), one lane per read ---------
        if (s == 0 && T > 0) {
            int leader_start = -1;
            int t = 0;
            while (t < T) {
                const int st = path[(size_t)t * 8 + rr];
                int e2 = t;
                while (e2 + 1 < T && path[(size_t)(e2 + 1) * 8 + rr] == st) e2++;
                if (st != P.adapter_state && st != P.lh_state && st != P.ll_state) {
                    leader_start = -1;
                } else {
                    if (leader_start < 0) leader_start = t;
                    if (st == P.adapter_state) {
                        const int64_t ev_last = k0 + e2, ev_lead = k0 + leader_start, ev_first = k0 + t;
                        const int64_t adapter_end = (ev_last == n_ev - 1)
                            ? first + (int64_t)P.stride * ev_last + 1
                            : first + (int64_t)P.stride * (ev_last + 1);
                        const int64_t leader_in_read = first + (int64_t)P.stride * ev_lead;
                        const int64_t total_duration = adapter_end - leader_in_read;
                        const int64_t adapter_duration = adapter_end - (first + (int64_t)P.stride * ev_first);
                        const int strict = (leader_in_read - payload_start) <= strict_duration ? 1 : 0;
                        if (total_duration >= cut_total[strict] && adapter_duration >= cut_adapter[strict]) {
                            if (count < PXG_MAX_UNSPLIT) {
                                out_iv[(r * PXG_MAX_UNSPLIT + count) * 2] = leader_in_read;
                                out_iv[(r * PXG_MAX_UNSPLIT + count) * 2 + 1] = 1 + adapter_end;
                            }
                            count++;
                        }
                        leader_start = -1;
                    }
                }
                t = e2 + 1;
            }
        }
        if (more) {
            left += window_step;
            more = left < last_end;
        }
    }
    if (s == 0 && r < n_reads) out_cnt[r] = overflow ? -1 : count;
}

int pxg_launch_unsplit_scan(pxg_ctx* ctx, int64_t n, const pxg_calib* cal, const int32_t* status,
                            const int32_t* segs, const int64_t* first_sample, const int64_t* ev_off,
                            const float* scaled, int stride, void* scratch, int64_t* out_iv,
                            int32_t* out_cnt)
{
    if (n <= 0) return PXG_OK;
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
    const size_t blocks = (size_t)(n + UN_READS - 1) / UN_READS;
    unsigned* bp = (unsigned*)scratch;
    unsigned char* path = (unsigned char*)scratch + blocks * (UN_TMAX / 8) * 64 * sizeof(unsigned);
    hipLaunchKernelGGL(k_unsplit_scan, dim3((unsigned)blocks), dim3(64), 0, ctx->stream, n, ctx->hmm[1], P,
                       cal, status, segs, first_sample, ev_off, scaled, bp, path, out_iv, out_cnt);
    return PXG_OK;
}

size_t pxg_unsplit_scratch_bytes(int64_t n)
{
    const size_t blocks = (size_t)(n + UN_READS - 1) / UN_READS;
    return blocks * ((UN_TMAX / 8) * 64 * sizeof(unsigned) + (size_t)UN_TMAX * 8);
}
