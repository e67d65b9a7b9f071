// k_lstm_q8_demux.hip -- K5a / K5b of the exact fixed-point arithmetic (k_lstm_q8.hip has the description, K2 and the
// host side).  A translation unit of its own because it is compiled with LLVM's "iterative-ilp" machine
// scheduler (Makefile): measured on one box, alternating, K5a 0.972 -> 0.889 ms and K5b 1.206 -> 1.016 ms against the
// default scheduler, which in turn is the best one for K2 (5.97 ms; iterative-ilp 6.10 with spills, max-ilp 6.45).
#include "k_lstm_q8.h"

// ===========================================================================
// K5a: demux bidirectional layer, time-sliced.  The two cells of a step are the forward net at time
// t and the backward net at time T-1-t.  Every step's digit planes (3 KB per direction and tile, the
// LDS image) are streamed to HBM as they are: bidir[tile][time][direction][Q8_HVEC] is exactly what
// K5b copies back into LDS as its B fragments.  Fragments: [0 forward U | 1 backward U][nt][digit][thread].
// ===========================================================================
#define Q8A_STATE (2 * Q8_HVEC / 4 + 2 * LSTM_THREADS * 3)      // dwords per saved tile state

__global__ __launch_bounds__(LSTM_THREADS, 2) void k_demux_bidir_q8(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const float* __restrict__ win, const float* __restrict__ sigtab, const v4i* __restrict__ frag,
    const float* __restrict__ Wf, const float* __restrict__ bf, const float* __restrict__ Wb,
    const float* __restrict__ bb, Q8Scale sf, Q8Scale sb, unsigned char* __restrict__ bidir,
    int* __restrict__ queue, int* __restrict__ errflag, int* __restrict__ done, unsigned* __restrict__ state)
{
    constexpr int H = 48, NT = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    const int n_tiles = (lim + 15) >> 4;
    const int n_blocks = queue[1] > 0 ? min(queue[1], T) : demux_blocks(n_tiles, (int)gridDim.x, T);
    const int QB = (T + n_blocks - 1) / n_blocks;
    const int n_tasks = n_tiles * n_blocks;

    float4* tab = reinterpret_cast<float4*>(smem);
    unsigned char* hv = reinterpret_cast<unsigned char*>(smem + 4 * PXG_SIG_NSEG);   // [2 buffers][fwd, bwd][Q8_HVEC]
    float* xf = reinterpret_cast<float*>(hv + 4 * Q8_HVEC);                    // [16][XS]  x[c0 + c]
    float* xr = xf + 16 * XS;                                                  // [16][XS]  x[T-1-(c0+c)]
    float4* gvec = reinterpret_cast<float4*>(xr + 16 * XS);                    // [4: bf, Wf, bb, Wb][48]
    int* ridx = reinterpret_cast<int*>(gvec + 4 * H);
    int* s_task = ridx + 16;

    const int tid = threadIdx.x, lane = tid & 63, slice = tid >> 6;
    const int rd_l = lane & 15, ul = lane >> 4;

    load_sigtab(tab, sigtab, tid);
    v4i wF[NT][3], wB[NT][3];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int d = 0; d < 3; d++) {
            wF[nt][d] = frag[((0 * NT + nt) * 3 + d) * LSTM_THREADS + tid];
            wB[nt][d] = frag[((1 * NT + nt) * 3 + d) * LSTM_THREADS + tid];
        }
    for (int i = tid; i < 4 * H; i += LSTM_THREADS) {
        const int v = i / H, unit = i % H;
        const float* src = v == 0 ? bf : (v == 1 ? Wf : (v == 2 ? bb : Wb));
        gvec[i] = make_float4(src[unit], src[H + unit], src[2 * H + unit], src[3 * H + unit]);
    }
    const int gv = slice * 12 + ul;
    constexpr int NST = (2 * Q8_HVEC / 16 + LSTM_THREADS - 1) / LSTM_THREADS;     // float4 per thread and step

    for (;;) {
        __syncthreads();
        if (tid == 0) *s_task = atomicAdd(&queue[0], 1);
        __syncthreads();
        const int q = *s_task;
        if (q >= n_tasks) break;
        const int blk = q / n_tiles, tile = q % n_tiles;
        const int t0 = blk * QB;
        const int t1 = min(t0 + QB, T);
        if (tid < 16) {
            const int row = tile * 16 + tid;
            ridx[tid] = row < lim ? (idx ? idx[row] : row) : -1;
        }
        float cf[NT], cb[NT];
        if (blk == 0) {
            for (int i = tid; i < 4 * Q8_HVEC / 4; i += LSTM_THREADS) reinterpret_cast<unsigned*>(hv)[i] = 0u;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) cf[nt] = cb[nt] = 0.0f;
            __syncthreads();
        } else {
            dq_wait(done, tile, blk, errflag, tid);
            const unsigned* st_in = state + ((size_t)((blk - 1) & 1) * n_tiles + tile) * Q8A_STATE;
            const int rb = t0 & 1;
            for (int i = tid; i < 2 * Q8_HVEC / 4; i += LSTM_THREADS)
                reinterpret_cast<unsigned*>(hv + rb * 2 * Q8_HVEC)[i] = st_in[i];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                cf[nt] = __uint_as_float(st_in[2 * Q8_HVEC / 4 + nt * LSTM_THREADS + tid]);
                cb[nt] = __uint_as_float(st_in[2 * Q8_HVEC / 4 + (NT + nt) * LSTM_THREADS + tid]);
            }
            __syncthreads();
        }
        unsigned char* tile_out = bidir + (size_t)tile * T * (2 * Q8_HVEC);

        for (int t = t0; t <= t1; t++) {
            const unsigned char* hr = hv + (t & 1) * 2 * Q8_HVEC;
            unsigned char* hw = hv + ((t + 1) & 1) * 2 * Q8_HVEC;
            if (t > t0) {    // stream the planes written in the previous step: forward time t-1, backward time T-t
#pragma unroll
                for (int p = 0; p < NST; p++) {
                    const int i = tid + p * LSTM_THREADS;
                    if (i < 2 * Q8_HVEC / 16) {
                        const int dir = i / (Q8_HVEC / 16), o = (i % (Q8_HVEC / 16)) * 16;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(hr + dir * Q8_HVEC + o);
                        const int time = dir ? T - t : t - 1;
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(tile_out + ((size_t)time * 2 + dir) * Q8_HVEC + o));
                    }
                }
            }
            if (t == t1) break;
            if (((t - t0) % XCH) == 0) {          // refill both x tiles
                __syncthreads();
                for (int i = tid; i < 16 * XCH; i += LSTM_THREADS) {
                    const int row = i / XCH, c = i % XCH;
                    const int rd = ridx[row];
                    const int tt = t + c;
                    const bool ok = rd >= 0 && tt < T;
                    xf[row * XS + c] = ok ? win[(size_t)rd * T + tt] : 0.0f;
                    xr[row * XS + c] = ok ? win[(size_t)rd * T + (T - 1 - tt)] : 0.0f;
                }
                __syncthreads();
            }
            v4i g1[3], g2[3];
            q8_load_b(g1, hr, lane);
            q8_load_b(g2, hr + Q8_HVEC, lane);
            const float x1 = xf[rd_l * XS + ((t - t0) % XCH)];
            const float x2 = xr[rd_l * XS + ((t - t0) % XCH)];
            f32x4 u[NT];
            {
                Q8Acc A[NT];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) q8_block<true>(A[nt], wF[nt][2], wF[nt][1], wF[nt][0], g1);
                f32x4 st[NT];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const float4 b = gvec[gv + nt * 4], w = gvec[H + gv + nt * 4];
                    st[nt][0] = __builtin_fmaf(x1, w.x, b.x); st[nt][1] = __builtin_fmaf(x1, w.y, b.y);
                    st[nt][2] = __builtin_fmaf(x1, w.z, b.z); st[nt][3] = __builtin_fmaf(x1, w.w, b.w);
                }
                q8_combine_tiles<NT>(u, A, sf, st);
            }
            Q8Acc B[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) q8_block<true>(B[nt], wB[nt][2], wB[nt][1], wB[nt][0], g2);
            {
                float hn[NT];
                unsigned plane[3];
                cells_update<NT>(tab, u, cf, hn);
                q8_pack<NT>(hn, plane);
                q8_store_h(hw, plane, slice, lane);
            }
            {
                f32x4 st[NT];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const float4 b = gvec[2 * H + gv + nt * 4], w = gvec[3 * H + gv + nt * 4];
                    st[nt][0] = __builtin_fmaf(x2, w.x, b.x); st[nt][1] = __builtin_fmaf(x2, w.y, b.y);
                    st[nt][2] = __builtin_fmaf(x2, w.z, b.z); st[nt][3] = __builtin_fmaf(x2, w.w, b.w);
                }
                q8_combine_tiles<NT>(u, B, sb, st);
                float hn[NT];
                unsigned plane[3];
                cells_update<NT>(tab, u, cb, hn);
                q8_pack<NT>(hn, plane);
                q8_store_h(hw + Q8_HVEC, plane, slice, lane);
            }
            __syncthreads();
        }

        if (t1 < T) {        // hand the tile over: state of iteration t1 -> HBM, then publish
            unsigned* st_out = state + ((size_t)(blk & 1) * n_tiles + tile) * Q8A_STATE;
            const int rb = t1 & 1;
            for (int i = tid; i < 2 * Q8_HVEC / 4; i += LSTM_THREADS)
                st_out[i] = reinterpret_cast<const unsigned*>(hv + rb * 2 * Q8_HVEC)[i];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                st_out[2 * Q8_HVEC / 4 + nt * LSTM_THREADS + tid] = __float_as_uint(cf[nt]);
                st_out[2 * Q8_HVEC / 4 + (NT + nt) * LSTM_THREADS + tid] = __float_as_uint(cb[nt]);
            }
            dq_publish(done, tile, blk, tid);
        }
    }
}

// ===========================================================================
// K5b: demux top cell (H = 64, inputs hf | hb of K5a) + Dense + softmax, time-sliced.  wave = 16 units
// (4 gate tiles), three 64-wide k blocks per gate tile: hf (48 used), hb (48 used), h3 (64).
// Fragments: [block 0 hf | 1 hb | 2 h3][nt][digit][thread]; the digit-0 fragments of the two input
// blocks live in LDS (144 weight registers would leave nothing for the accumulators).
// ===========================================================================
#define Q8B_STATE (Q8_HVEC / 4 + LSTM_THREADS * 4)             // dwords per saved tile state

__global__ __launch_bounds__(LSTM_THREADS, 2) void k_demux_top_q8(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const unsigned char* __restrict__ bidir, const float* __restrict__ sigtab, const v4i* __restrict__ frag,
    const float* __restrict__ b3, Q8Scale s3, const float* __restrict__ Wd, const float* __restrict__ bd,
    int n_classes, float* __restrict__ probs, int* __restrict__ queue, int* __restrict__ errflag,
    int* __restrict__ done, unsigned* __restrict__ state)
{
    constexpr int H = 64, NT = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    const int n_tiles = (lim + 15) >> 4;
    const int n_blocks = queue[1] > 0 ? min(queue[1], T) : demux_blocks(n_tiles, (int)gridDim.x, T);
    const int QB = (T + n_blocks - 1) / n_blocks;
    const int n_tasks = n_tiles * n_blocks;

    float4* tab = reinterpret_cast<float4*>(smem);
    unsigned char* h3 = reinterpret_cast<unsigned char*>(smem + 4 * PXG_SIG_NSEG);   // [2 buffers][Q8_HVEC]
    unsigned char* inb = h3 + 2 * Q8_HVEC;                                     // [2 buffers][hf, hb][Q8_HVEC]
    v4i* w0s = reinterpret_cast<v4i*>(inb + 4 * Q8_HVEC);                      // [2 blocks][NT][thread]
    float4* gvec = reinterpret_cast<float4*>(w0s + 2 * NT * LSTM_THREADS);     // [64] bias (i, f, g, o)
    int* ridx = reinterpret_cast<int*>(gvec + H);
    int* s_task = ridx + 16;

    const int tid = threadIdx.x, lane = tid & 63, slice = tid >> 6;
    const int ul = lane >> 4;

    load_sigtab(tab, sigtab, tid);
    v4i wI[2][NT][2], wR[NT][3];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            wI[b][nt][0] = frag[((b * NT + nt) * 3 + 1) * LSTM_THREADS + tid];
            wI[b][nt][1] = frag[((b * NT + nt) * 3 + 2) * LSTM_THREADS + tid];
            w0s[(b * NT + nt) * LSTM_THREADS + tid] = frag[((b * NT + nt) * 3 + 0) * LSTM_THREADS + tid];
        }
#pragma unroll
        for (int d = 0; d < 3; d++) wR[nt][d] = frag[((2 * NT + nt) * 3 + d) * LSTM_THREADS + tid];
    }
    for (int i = tid; i < H; i += LSTM_THREADS) gvec[i] = make_float4(b3[i], b3[H + i], b3[2 * H + i], b3[3 * H + i]);
    const int gv = slice * 16 + ul;
    constexpr int NPF = (2 * Q8_HVEC / 16 + LSTM_THREADS - 1) / LSTM_THREADS;

    for (;;) {
        __syncthreads();
        if (tid == 0) *s_task = atomicAdd(&queue[0], 1);
        __syncthreads();
        const int q = *s_task;
        if (q >= n_tasks) break;
        const int blk = q / n_tiles, tile = q % n_tiles;
        const int t0 = blk * QB;
        const int t1 = min(t0 + QB, T);
        if (tid < 16) {
            const int row = tile * 16 + tid;
            ridx[tid] = row < lim ? (idx ? idx[row] : row) : -1;
        }
        float c3[NT];
        if (blk == 0) {
            for (int i = tid; i < 2 * Q8_HVEC / 4; i += LSTM_THREADS) reinterpret_cast<unsigned*>(h3)[i] = 0u;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) c3[nt] = 0.0f;
        } else {
            dq_wait(done, tile, blk, errflag, tid);
            const unsigned* st_in = state + ((size_t)((blk - 1) & 1) * n_tiles + tile) * Q8B_STATE;
            const int rb = t0 & 1;
            for (int i = tid; i < Q8_HVEC / 4; i += LSTM_THREADS) reinterpret_cast<unsigned*>(h3 + rb * Q8_HVEC)[i] = st_in[i];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) c3[nt] = __uint_as_float(st_in[Q8_HVEC / 4 + nt * LSTM_THREADS + tid]);
        }
        // input planes of step t0 into the buffer step t0 reads
        const unsigned char* tile_in = bidir + (size_t)tile * T * (2 * Q8_HVEC);
#pragma unroll
        for (int p = 0; p < NPF; p++) {
            const int i = tid + p * LSTM_THREADS;
            if (i < 2 * Q8_HVEC / 16)
                *reinterpret_cast<f32x4*>(inb + (t0 & 1) * 2 * Q8_HVEC + i * 16) =
                    *reinterpret_cast<const f32x4*>(tile_in + (size_t)t0 * 2 * Q8_HVEC + i * 16);
        }
        __syncthreads();

        for (int t = t0; t < t1; t++) {
            const int rdb = t & 1, wrb = (t + 1) & 1;
            f32x4 pf[NPF];
#pragma unroll
            for (int p = 0; p < NPF; p++) {
                const int i = tid + p * LSTM_THREADS;
                pf[p] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i < 2 * Q8_HVEC / 16 && t + 1 < t1)
                    pf[p] = *reinterpret_cast<const f32x4*>(tile_in + (size_t)(t + 1) * 2 * Q8_HVEC + i * 16);
            }
            Q8Acc A[NT];
            {
                v4i g[3];
                q8_load_b(g, inb + rdb * 2 * Q8_HVEC, lane);
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    q8_block<true>(A[nt], wI[0][nt][1], wI[0][nt][0], w0s[(0 * NT + nt) * LSTM_THREADS + tid], g);
                q8_load_b(g, inb + rdb * 2 * Q8_HVEC + Q8_HVEC, lane);
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    q8_block<false>(A[nt], wI[1][nt][1], wI[1][nt][0], w0s[(1 * NT + nt) * LSTM_THREADS + tid], g);
                q8_load_b(g, h3 + rdb * Q8_HVEC, lane);
#pragma unroll
                for (int nt = 0; nt < NT; nt++) q8_block<false>(A[nt], wR[nt][2], wR[nt][1], wR[nt][0], g);
            }
            float hn[NT];
#pragma unroll
            for (int half = 0; half < 2; half++) {      // 2 x 2 cells: 8 table rows in flight at a time
                f32x4 u2[2], st[2];
                const Q8Acc A2[2] = { A[2 * half], A[2 * half + 1] };
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const float4 b = gvec[gv + (2 * half + k) * 4];
                    st[k][0] = b.x; st[k][1] = b.y; st[k][2] = b.z; st[k][3] = b.w;
                }
                q8_combine_tiles<2>(u2, A2, s3, st);
                float cc[2] = { c3[2 * half], c3[2 * half + 1] };
                float hh[2];
                cells_update<2>(tab, u2, cc, hh);
                c3[2 * half] = cc[0]; c3[2 * half + 1] = cc[1];
                hn[2 * half] = hh[0]; hn[2 * half + 1] = hh[1];
            }
            unsigned plane[3];
            q8_pack<NT>(hn, plane);
            q8_store_h(h3 + wrb * Q8_HVEC, plane, slice, lane);
#pragma unroll
            for (int p = 0; p < NPF; p++) {
                const int i = tid + p * LSTM_THREADS;
                if (i < 2 * Q8_HVEC / 16) *reinterpret_cast<f32x4*>(inb + wrb * 2 * Q8_HVEC + i * 16) = pf[p];
            }
            __syncthreads();
        }

        if (t1 == T) {
            // ---- Dense(n_classes) + softmax on h = q * 2^-22 ---------------------------
            const unsigned char* hf = h3 + (T & 1) * Q8_HVEC;
            for (int row = tid; row < 16; row += LSTM_THREADS) {
                const int rd = ridx[row];
                if (rd < 0) continue;
                float z[PXG_MAX_CLASSES], e[PXG_MAX_CLASSES];
#pragma unroll
                for (int j = 0; j < PXG_MAX_CLASSES; j++) {
                    z[j] = -__builtin_inff();
                    if (j < n_classes) {
                        float acc = bd[j];
                        for (int k = 0; k < H; k++) acc = __builtin_fmaf(q8_read_h<16>(hf, row, k), Wd[k * n_classes + j], acc);
                        z[j] = acc;
                    }
                }
                float mx = z[0];
#pragma unroll
                for (int j = 1; j < PXG_MAX_CLASSES; j++) mx = (j < n_classes && z[j] > mx) ? z[j] : mx;
                float sden = 0.0f;
#pragma unroll
                for (int j = 0; j < PXG_MAX_CLASSES; j++) {
                    e[j] = j < n_classes ? pxg_expf(z[j] - mx) : 0.0f;
                    if (j < n_classes) sden = (j == 0) ? e[0] : sden + e[j];
                }
#pragma unroll
                for (int j = 0; j < PXG_MAX_CLASSES; j++)
                    probs[(size_t)rd * PXG_MAX_CLASSES + j] = j < n_classes ? e[j] / sden : 0.0f;
            }
        } else {
            unsigned* st_out = state + ((size_t)(blk & 1) * n_tiles + tile) * Q8B_STATE;
            const int rb = t1 & 1;
            for (int i = tid; i < Q8_HVEC / 4; i += LSTM_THREADS) st_out[i] = reinterpret_cast<const unsigned*>(h3 + rb * Q8_HVEC)[i];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) st_out[Q8_HVEC / 4 + nt * LSTM_THREADS + tid] = __float_as_uint(c3[nt]);
            dq_publish(done, tile, blk, tid);
        }
    }
}


// bidir: at least ceil(n_rows / 16) * T * 2 * Q8_HVEC bytes (= the float32 layout's 96 floats per read
// and step, with the row count rounded up to whole tiles)
int pxg_launch_demux_lstm_q8(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx, const int32_t* count,
                             const float* win, float* bidir, float* probs, int timer_a, int timer_b)
{
    if (n_rows <= 0) return PXG_OK;
    {   // small batches: 4-read tiles on four times as many CUs (k_lstm_q8_lat.hip); PXG_K5_LAT_MAX = 0 switches it off
        const char* lat_env = getenv("PXG_K5_LAT_MAX");
        const int64_t lat_max = lat_env ? atoll(lat_env) : 8 * (int64_t)ctx->n_cu;
        if (n_rows <= lat_max) return pxg_launch_demux_lstm_q8_lat(ctx, n_rows, idx, count, win, bidir, probs, timer_a, timer_b);
    }
    const int T = ctx->cfg.signal_trim_length;
    const int64_t tiles = (n_rows + 15) / 16, slots = 2 * (int64_t)ctx->n_cu;
    const int64_t grid = std::min(tiles, slots);
    const size_t state_dw = (size_t)2 * tiles * (Q8A_STATE > Q8B_STATE ? Q8A_STATE : Q8B_STATE);
    int rc;
    if (!ctx->lstm_err.p) {
        if ((rc = pxg_reserve(ctx, ctx->lstm_err, 1))) return rc;
        PXG_HIP(ctx, hipMemsetAsync(ctx->lstm_err.p, 0, sizeof(int), ctx->stream));
    }
    if ((rc = pxg_reserve(ctx, ctx->demux_q, (size_t)2 * (2 + tiles))) || (rc = pxg_reserve(ctx, ctx->demux_state, state_dw)))
        return rc;
    PXG_HIP(ctx, hipMemsetAsync(ctx->demux_q.p, 0, (size_t)2 * (2 + tiles) * sizeof(int), ctx->stream));
    int* qa = ctx->demux_q.p;
    int* qb = ctx->demux_q.p + 2 + tiles;
    if (const char* forced = getenv("PXG_DEMUX_BLOCKS")) {      // tuning knob: step blocks per tile ("a,b" or one number)
        static int nb_a, nb_b;
        nb_a = nb_b = atoi(forced);
        for (const char* c = forced; *c; c++)
            if (*c == ',') nb_b = atoi(c + 1);
        PXG_HIP(ctx, hipMemcpyAsync(qa + 1, &nb_a, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        PXG_HIP(ctx, hipMemcpyAsync(qb + 1, &nb_b, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    }
    const PxgLstmDev &f = ctx->demux_fwd, &b = ctx->demux_bwd, &t3 = ctx->demux_top;
    unsigned* st = reinterpret_cast<unsigned*>(ctx->demux_state.p);
    {
        const size_t lds = sizeof(float) * 4 * PXG_SIG_NSEG + 4 * Q8_HVEC + sizeof(float) * 2 * 16 * XS + 16 * 4 * 48 + sizeof(int) * 32;
        PXG_HIP(ctx, hipFuncSetAttribute((const void*)k_demux_bidir_q8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        pxg_timer_begin(ctx, timer_a);
        hipLaunchKernelGGL(k_demux_bidir_q8, dim3((unsigned)grid), dim3(LSTM_THREADS), lds, ctx->stream, (int)n_rows, idx,
                           count, T, win, ctx->d_sigtab, reinterpret_cast<const v4i*>(ctx->q8.bidir_frag), f.kernel, f.bias,
                           b.kernel, b.bias, q8_scale(ctx->q8.s_fwd), q8_scale(ctx->q8.s_bwd), reinterpret_cast<unsigned char*>(bidir), qa,
                           ctx->lstm_err.p, qa + 2, st);
        pxg_timer_end(ctx, timer_a);
    }
    {
        const size_t lds = sizeof(float) * 4 * PXG_SIG_NSEG + 2 * Q8_HVEC + 4 * Q8_HVEC + 16 * 2 * 4 * LSTM_THREADS + 16 * 64 +
                           sizeof(int) * 32;
        PXG_HIP(ctx, hipFuncSetAttribute((const void*)k_demux_top_q8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        pxg_timer_begin(ctx, timer_b);
        hipLaunchKernelGGL(k_demux_top_q8, dim3((unsigned)grid), dim3(LSTM_THREADS), lds, ctx->stream, (int)n_rows, idx, count,
                           T, reinterpret_cast<const unsigned char*>(bidir), ctx->d_sigtab,
                           reinterpret_cast<const v4i*>(ctx->q8.top_frag), t3.bias, q8_scale(ctx->q8.s_top), ctx->demux_dense.kernel,
                           ctx->demux_dense.bias, ctx->demux_dense.out_dim, probs, qb, ctx->lstm_err.p, qb + 2, st);
        pxg_timer_end(ctx, timer_b);
    }
    ctx->timeslice_used = true;
    PXG_HIP(ctx, hipGetLastError());
    return PXG_OK;
}
