// k_lstm_q8.hip -- K2 / K5 with the recurrent matrix products in EXACT fixed point on the int8
// matrix pipe (pxg_config.lstm_arith == PXG_LSTM_Q8; include/pxg.h, DESIGN.md 3.1 "round 4").
//
// Why: on gfx950 v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate and shares the FP32 lanes with
// the gate math (rounds 1-3: K2 pinned at 0.68 of that roof), while v_mfma_i32_16x16x64_i8 retires
// 16x the k per instruction in ~2/3 of the cycles on a pipe the VALU does not use.  A float32 split
// (bf16 hi/mid/lo on the bf16 pipe) would make the result depend on the order in which the hardware
// adds 32 products; integer sums do not depend on any order, so the arithmetic below is exact,
// restated bit for bit by oracle/pxo_core.c (lstm_step_q) and checked against the integer dot
// product on the device (tools/ubench/i8_mfma.hip).
//
// Arithmetic (the oracle's comment block is the specification):
//   h = q * 2^-22, q = rint(fl(o * tanh(c')) * 2^22);  Wq = rint(W * 2^p), p per layer;
//   balanced base-256 digits h0..h2, w0..w2 (each in [-128, 127]);
//   A0 = sum w2 h2, A1 = sum w2 h1 + w1 h2, A2 = sum w2 h0 + w1 h1 + w0 h2, A3 = sum w1 h0 + w0 h1
//   (eight MFMAs per 64-wide k block and 16x16 gate tile, int32 accumulators, exact);
//   V = A0 * 256 + A1, U = A2 * 256 + A3, t = fma((float)V, 65536, (float)U),
//   u = fma(t, g * 2^(-p-14), start): the pre-activation in table units; gates / cell as in k_lstm.hip.
//
// Layout.  A 4-wave workgroup owns 16 reads; wave w owns units [12w, 12w + 12) (scaler, K5a) or
// [16w, 16w + 16) (K5b): NT gate tiles of 4 units x 4 gates, the four gates of one (read, unit) in
// the four accumulator registers of one lane (as in k_lstm.hip).  The k index of a unit is chosen so
// that the NT cells a lane updates are NT consecutive bytes: kk = 16 w + 4 (lane >> 4) + nt -- a
// lane publishes its cells' digits with ONE ds_write_b32 per digit plane, a wave's stores are 256
// contiguous bytes, and a 48-unit layer fills a 64-wide block with every fourth byte zero (its
// weights there are zero as well).  Digit plane d of a hidden vector in LDS: [k group (4)][read
// (16)][16 bytes]; the B fragment of lane (read = lane & 15, group = lane >> 4) is one
// ds_read_b128, conflict-free.  Weight fragments are laid out by the host in exactly the register
// order (pxg_q8_upload) and stay in VGPRs for the whole launch.
#include "k_lstm_q8.h"


// ===========================================================================
// K2: scaler, time-sliced (tasks = (step block, tile) as in k_scaler_lstm_q; with no more tiles
// than slots every tile is one task).
//
// Round 5: layer 2's k extent without padding.  Layer 2 multiplies [h1(t-1) | h2(t-2)] = 96 units; as two 64-wide
// blocks with every fourth byte zero that was 16 MFMAs per gate tile.  Now the two hidden vectors of a read share ONE
// 96-byte row per digit plane:
//   block 0 (k groups 0-3, 64 bytes): lane (wave w, ul) owns bytes 4 ul .. 4 ul + 3 of group w =
//            its three layer-1 cells and its FIRST layer-2 cell -- one ds_write_b32 per plane, as before;
//   block 1' (k groups 4-5, 32 bytes): its other two layer-2 cells, bytes 2 (4 w + ul), + 1 -- one ds_write_b16.
// Layer 1 takes block 0 with zero weights at the layer-2 bytes (8 MFMAs per tile, as before: a 48-unit layer cannot
// fill 64-wide blocks level by level).  Layer 2 takes block 0 whole (8 MFMAs, the SAME B fragments) and block 1' as
// HALF blocks paired by significance level -- an MFMA may sum any products of one level:
//   F1 = [h1' | h2'], F2 = [h0' | h1']  (digit planes of block 1'; per lane ONE ds_read_b128 at a plane / group that
//                                        depends on the lane's k group)
//   A0 += [ 0 | w2'] F1      A1 += [w2' | w1'] F1      A2 += [w2' | w1'] F2 + [ 0 | w0'] F1      A3 += [w1' | w0'] F2
// 5 MFMAs instead of 8: 13 per layer-2 tile, 63 per wave and step instead of 72, 5 B fragments instead of 6.  The
// sums are the same integers (the oracle does not know about blocks).
// Fragments: [nt][10 ids][thread]: 0-2 layer 1 (w2, w1, w0 over block 0), 3-5 layer 2 over block 0, 6 [w2'|w1'],
// 7 [w1'|w0'], 8 [0|w2'], 9 [0|w0'].
// ===========================================================================
#define Q2_PLANE (6 * 256)                  // one digit plane: [6 k groups][16 reads][16 bytes]
#define Q2_HVEC (3 * Q2_PLANE)              // both hidden vectors of a tile's 16 reads
#define Q2_FRAGS 10
#define Q2_LDS_FRAGS 4                      // ids 2, 5, 7, 9 are read from LDS where they are used
#define Q8S_STATE (Q2_HVEC / 4 + 2 * LSTM_THREADS * 3)      // dwords per saved tile state
#define Q8S_TRAJ 4          // the zero-input trajectory is kept every 4 steps (step blocks are multiples of 4)

// h (float, exact) of layer-2 unit `unit` (wave-major numbering, 12 units per wave) of read `rd`
__device__ __forceinline__ float q2_read_h2(const unsigned char* hvec, int rd, int unit)
{
    const int w = unit / 12, nt = (unit % 12) >> 2, ul = unit & 3;
    const int off = nt == 0 ? (((w * 16 + rd) << 4) + (ul << 2) + 3)
                            : ((((4 + (w >> 1)) * 16 + rd) << 4) + 2 * (4 * (w & 1) + ul) + (nt - 1));
    const int q = (int)(signed char)hvec[off] + 256 * (int)(signed char)hvec[Q2_PLANE + off] +
                  65536 * (int)(signed char)hvec[2 * Q2_PLANE + off];
    return (float)q * (1.0f / 4194304.0f);
}

__global__ __launch_bounds__(LSTM_THREADS, 2) void k_scaler_lstm_q8(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const float* __restrict__ head, const float* __restrict__ sigtab, const v4i* __restrict__ frag,
    const float* __restrict__ W1 /* scalar-input kernel, table units */, const float* __restrict__ b1,
    const float* __restrict__ b2, Q8Scale s1, Q8Scale s2 /* g * 2^(-p-14) of the gate blocks of layer 1 / 2 */,
    const float* __restrict__ Wd, const float* __restrict__ bd, float* __restrict__ pred,
    int* __restrict__ queue, int* __restrict__ errflag, int* __restrict__ done, unsigned* __restrict__ state,
    const int64_t* __restrict__ off /* sample offsets of the reads, or null: no read is known to be padded */,
    int head_limit, int stride, const unsigned* __restrict__ traj /* zero-input states every Q8S_TRAJ steps, or null */,
    unsigned* __restrict__ traj_out /* not null: this launch records that table */)
{
    constexpr int H = 48, NT = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    const int n_tiles = (lim + 15) >> 4;
    const int QBS = queue[1] > 0 ? ((min(queue[1], T + 1) + 3) & ~3) : scaler_block_steps(n_tiles, (int)gridDim.x, T + 1);
    const int n_blocks = (T + 1 + QBS - 1) / QBS;
    const int n_tasks = n_tiles * n_blocks;

    float4* tab = reinterpret_cast<float4*>(smem);                             // [1024] sigmoid spline
    unsigned char* hv = reinterpret_cast<unsigned char*>(smem + 4 * PXG_SIG_NSEG);   // [2 buffers][Q2_HVEC]
    float* xb = reinterpret_cast<float*>(hv + 2 * Q2_HVEC);                    // [16][XS]
    float4* gvec = reinterpret_cast<float4*>(xb + 16 * XS);                    // [3: b1, W1, b2][4 waves][NT][4 ul]
    v4i* wl = reinterpret_cast<v4i*>(gvec + 3 * 48);                           // [NT][Q2_LDS_FRAGS][thread]
    int* ridx = reinterpret_cast<int*>(wl + NT * Q2_LDS_FRAGS * LSTM_THREADS); // [16]
    int* s_task = ridx + 16;

    const int tid = threadIdx.x, lane = tid & 63, slice = tid >> 6;
    const int rd_l = lane & 15, ul = lane >> 4;

    load_sigtab(tab, sigtab, tid);
    // weight fragments of this wave's gate tiles: six per tile in VGPRs for the whole launch (72 registers, as
    // before), the four that carry digit 0 in LDS -- with all ten in registers nothing is left for the accumulators
    // and the table rows a cell update keeps in flight
    v4i wA[NT][2], wB[NT][2], wH[NT][2];       // [w2, w1] of layer 1 / layer 2 over block 0; [w2'|w1'], [0|w2']
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const v4i* f = frag + (size_t)nt * Q2_FRAGS * LSTM_THREADS + tid;
        wA[nt][0] = f[0 * LSTM_THREADS]; wA[nt][1] = f[1 * LSTM_THREADS];
        wB[nt][0] = f[3 * LSTM_THREADS]; wB[nt][1] = f[4 * LSTM_THREADS];
        wH[nt][0] = f[6 * LSTM_THREADS]; wH[nt][1] = f[8 * LSTM_THREADS];
        wl[(nt * Q2_LDS_FRAGS + 0) * LSTM_THREADS + tid] = f[2 * LSTM_THREADS];
        wl[(nt * Q2_LDS_FRAGS + 1) * LSTM_THREADS + tid] = f[5 * LSTM_THREADS];
        wl[(nt * Q2_LDS_FRAGS + 2) * LSTM_THREADS + tid] = f[7 * LSTM_THREADS];
        wl[(nt * Q2_LDS_FRAGS + 3) * LSTM_THREADS + tid] = f[9 * LSTM_THREADS];
    }
    for (int i = tid; i < 3 * H; i += LSTM_THREADS) {      // (i, f, g, o) of one unit per float4
        const int v = i / H, unit = i % H;
        const float* src = v == 0 ? b1 : (v == 1 ? W1 : b2);
        gvec[i] = make_float4(src[unit], src[H + unit], src[2 * H + unit], src[3 * H + unit]);
    }
    const int gv = slice * 12 + ul;                        // + nt * 4: this lane's units
    // byte offsets inside a hidden-vector buffer: the block-0 fragment of plane 0, the two half-block fragments
    // (their plane depends on the lane's k group), this lane's dword of block 0 and its half word of block 1'
    const int off_g = ((ul * 16 + rd_l) << 4);
    const int off_h = (((4 + (ul & 1)) * 16 + rd_l) << 4);
    const int off_f1 = (ul < 2 ? 1 : 2) * Q2_PLANE + off_h;
    const int off_f2 = (ul < 2 ? 0 : 1) * Q2_PLANE + off_h;
    const int off_w0 = ((slice * 16 + rd_l) << 4) + (ul << 2);
    const int off_w1 = (((4 + (slice >> 1)) * 16 + rd_l) << 4) + 2 * (4 * (slice & 1) + ul);

    for (;;) {
        __syncthreads();                       // everybody is done with the previous task's LDS
        if (tid == 0) *s_task = atomicAdd(&queue[0], 1);
        __syncthreads();
        const int q = *s_task;
        if (q >= n_tasks) break;
        const int blk = q / n_tiles, tile = q % n_tiles;
        const int t0 = blk * QBS;
        const int t1 = min(t0 + QBS, T + 1);
        const int row_base = tile * 16;
        if (tid < 16) {
            const int row = row_base + tid;
            const int rd = row < lim ? (idx ? idx[row] : row) : -1;
            ridx[tid] = rd;
            // The head is LEFT-padded with zeros (signal_loader.py:227-229; K1): until its first sample a read
            // is the zero-input network started from zero, the same for every read.  A tile whose 16 reads
            // all start at or after step b0 * QBS takes that state from the table and skips blocks < b0.
            int pad = T + 1;
            if (rd >= 0) {
                pad = 0;
                if (off && traj) {
                    const int64_t len = off[rd + 1] - off[rd];
                    pad = T - (int)((len < head_limit ? len : head_limit) / stride);
                }
            }
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) pad = min(pad, __shfl_xor(pad, d));
            if (tid == 0) s_task[1] = min(max(pad, 0) / QBS, n_blocks - 1);
        }
        __syncthreads();
        const int b0 = s_task[1];
        if (blk < b0) continue;            // nothing to do and nobody waits for it
        float c1[NT], c2[NT];
        // (a recording launch hands over THROUGH the table: one tile, its blocks in order)
        const unsigned* st_in = traj_out ? traj_out + (size_t)(t0 / Q8S_TRAJ) * Q8S_STATE
                                : blk == b0 ? traj + (size_t)(t0 / Q8S_TRAJ) * Q8S_STATE
                                            : state + ((size_t)((blk - 1) & 1) * n_tiles + tile) * Q8S_STATE;
        if (blk == 0) {
            for (int i = tid; i < 2 * Q2_HVEC / 4; i += LSTM_THREADS) reinterpret_cast<unsigned*>(hv)[i] = 0u;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) c1[nt] = c2[nt] = 0.0f;
        } else {
            if (blk != b0) dq_wait(done, tile, blk, errflag, tid);
            const int rb = t0 & 1;         // (the other buffer is written completely by step t0)
            for (int i = tid; i < Q2_HVEC / 4; i += LSTM_THREADS)
                reinterpret_cast<unsigned*>(hv + rb * Q2_HVEC)[i] = st_in[i];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                c1[nt] = __uint_as_float(st_in[Q2_HVEC / 4 + nt * LSTM_THREADS + tid]);
                c2[nt] = __uint_as_float(st_in[Q2_HVEC / 4 + (NT + nt) * LSTM_THREADS + tid]);
            }
        }
        __syncthreads();

        for (int t = t0; t < t1; t++) {
            if (((t - t0) % XCH) == 0 && t < T) {        // refill the x tile (rows x XCH steps)
                __syncthreads();
                for (int i = tid; i < 16 * (XCH / 4); i += LSTM_THREADS) {
                    const int row = i / (XCH / 4), c4 = i % (XCH / 4);
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int rd = ridx[row];
                    if (rd >= 0 && t + c4 * 4 < T)
                        v = *reinterpret_cast<const float4*>(head + (size_t)rd * T + t + c4 * 4);
                    *reinterpret_cast<float4*>(xb + row * XS + c4 * 4) = v;
                }
                __syncthreads();
            }
            const unsigned char* hr = hv + (t & 1) * Q2_HVEC;
            unsigned char* hw = hv + ((t + 1) & 1) * Q2_HVEC;
            v4i g[3];                                  // block 0 of the three planes: h1(t-1) and a quarter of h2(t-2)
#pragma unroll
            for (int d = 0; d < 3; d++) g[d] = *reinterpret_cast<const v4i*>(hr + d * Q2_PLANE + off_g);
            const v4i f1 = *reinterpret_cast<const v4i*>(hr + off_f1);      // [h1' | h2'] of block 1'
            const v4i f2 = *reinterpret_cast<const v4i*>(hr + off_f2);      // [h0' | h1']
            const float x = xb[rd_l * XS + ((t - t0) % XCH)];

            // ---- layer 1, step t ---------------------------------------------------------
            f32x4 u[NT];
            {
                Q8Acc A[NT];
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    q8_block<true>(A[nt], wA[nt][0], wA[nt][1], wl[(nt * Q2_LDS_FRAGS + 0) * LSTM_THREADS + tid], g);
                f32x4 st[NT];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const float4 b = gvec[gv + nt * 4], w = gvec[H + gv + nt * 4];
                    st[nt][0] = __builtin_fmaf(x, w.x, b.x); st[nt][1] = __builtin_fmaf(x, w.y, b.y);
                    st[nt][2] = __builtin_fmaf(x, w.z, b.z); st[nt][3] = __builtin_fmaf(x, w.w, b.w);
                }
                q8_combine_tiles<NT>(u, A, s1, st);
            }
            // ---- layer 2, step t - 1: block 0 whole, block 1' as level-paired halves; layer 1's cells under its MFMAs
            Q8Acc B[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                q8_block<true>(B[nt], wB[nt][0], wB[nt][1], wl[(nt * Q2_LDS_FRAGS + 1) * LSTM_THREADS + tid], g);
                B[nt].a0 = mfma8(wH[nt][1], f1, B[nt].a0);                                              // [ 0 | w2'] [h1' | h2']
                B[nt].a1 = mfma8(wH[nt][0], f1, B[nt].a1);                                              // [w2'|w1'] [h1' | h2']
                B[nt].a2 = mfma8(wH[nt][0], f2, B[nt].a2);                                              // [w2'|w1'] [h0' | h1']
                B[nt].a3 = mfma8(wl[(nt * Q2_LDS_FRAGS + 2) * LSTM_THREADS + tid], f2, B[nt].a3);       // [w1'|w0'] [h0' | h1']
                B[nt].a2 = mfma8(wl[(nt * Q2_LDS_FRAGS + 3) * LSTM_THREADS + tid], f1, B[nt].a2);       // [ 0 | w0'] [h1' | h2']
            }
            unsigned plane1[3] = {0u, 0u, 0u}, plane2[3] = {0u, 0u, 0u};
            if (t < T) {
                float hn[NT];
                cells_update<NT>(tab, u, c1, hn);
                q8_pack<NT>(hn, plane1);
            }
            {
                f32x4 st[NT];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const float4 b = gvec[2 * H + gv + nt * 4];
                    st[nt][0] = b.x; st[nt][1] = b.y; st[nt][2] = b.z; st[nt][3] = b.w;
                }
                q8_combine_tiles<NT>(u, B, s2, st);
            }
            if (t >= 1) {
                float hn[NT];
                cells_update<NT>(tab, u, c2, hn);
                q8_pack<NT>(hn, plane2);
            }
            // this lane's bytes of the next row: three layer-1 digits + the first layer-2 digit as one dword of
            // block 0, the other two layer-2 digits as one half word of block 1'
#pragma unroll
            for (int d = 0; d < 3; d++) {
                *reinterpret_cast<unsigned*>(hw + d * Q2_PLANE + off_w0) = plane1[d] | (plane2[d] << 24);
                *reinterpret_cast<unsigned short*>(hw + d * Q2_PLANE + off_w1) = (unsigned short)(plane2[d] >> 8);
            }
            __syncthreads();
        }

        if (t1 == T + 1) {
            // ---- Dense(2): fma chain over k = 0..47 from the bias, on h = q * 2^-22 ----------
            const unsigned char* hf = hv + ((T + 1) & 1) * Q2_HVEC;
            for (int i = tid; i < 16 * 2; i += LSTM_THREADS) {
                const int row = i >> 1, j = i & 1;
                const int rd = ridx[row];
                if (rd < 0) continue;
                float acc = bd[j];
                for (int k = 0; k < H; k++) acc = __builtin_fmaf(q2_read_h2(hf, row, k), Wd[k * 2 + j], acc);
                pred[(size_t)rd * 2 + j] = acc;
            }
        } else {
            // ---- hand the tile over: state of iteration t1 -> HBM, then publish -------------
            unsigned* st_out = traj_out ? traj_out + (size_t)(t1 / Q8S_TRAJ) * Q8S_STATE
                                        : state + ((size_t)(blk & 1) * n_tiles + tile) * Q8S_STATE;
            const int rb = t1 & 1;
            for (int i = tid; i < Q2_HVEC / 4; i += LSTM_THREADS)
                st_out[i] = reinterpret_cast<const unsigned*>(hv + rb * Q2_HVEC)[i];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                st_out[Q2_HVEC / 4 + nt * LSTM_THREADS + tid] = __float_as_uint(c1[nt]);
                st_out[Q2_HVEC / 4 + (NT + nt) * LSTM_THREADS + tid] = __float_as_uint(c2[nt]);
            }
            dq_publish(done, tile, blk, tid);
        }
    }
}


// ===========================================================================
// host side: integer matrices, digit planes in fragment order
// ===========================================================================
namespace {

struct Q8Mat {
    int K = 0, G = 0, p[4] = {0, 0, 0, 0};
    std::vector<int32_t> w;        // [K][G]
    void scales(float (&s)[4]) const { for (int g = 0; g < 4; g++) s[g] = (float)std::ldexp(g == 2 ? 32.0 : 16.0, -p[g] - 14); }
};

// rows = the vector-input rows (input_dim > 1) then the recurrent rows; one exponent per gate block (i, f, c, o)
Q8Mat q8_quantise(const pxg_lstm_layer& L)
{
    const int H = L.units, G = 4 * H, Kin = L.input_dim == 1 ? 0 : L.input_dim;
    Q8Mat M;
    M.K = Kin + H;
    M.G = G;
    auto at = [&](int k, int j) { return (double)(k < Kin ? L.kernel[(size_t)k * G + j] : L.recurrent[(size_t)(k - Kin) * G + j]); };
    for (int g = 0; g < 4; g++) {
        double m = 0.0;
        for (int k = 0; k < M.K; k++)
            for (int j = g * H; j < (g + 1) * H; j++) m = std::max(m, std::fabs(at(k, j)));
        int p = 0;
        if (m > 0.0)
            for (p = 40; std::ldexp(m, p) > 8355711.0; p--) {}
        M.p[g] = p;
    }
    M.w.resize((size_t)M.K * G);
    for (int k = 0; k < M.K; k++)
        for (int j = 0; j < G; j++) M.w[(size_t)k * G + j] = (int32_t)std::nearbyint(std::ldexp(at(k, j), M.p[j / H]));
    return M;
}

inline int q8_digit(int32_t v, int d)
{
    const uint32_t b = (uint32_t)(v + 0x808080);          // bytes of v + 0x808080, each - 128
    return (int)((b >> (8 * d)) & 255u) - 128;
}

// k position -> unit of a hidden vector written by a kernel with UPW units per wave (12: every
// fourth byte unused; 16: all used)
inline int q8_unit_of_k(int kk, int upw)
{
    const int w = kk >> 4, ul = (kk >> 2) & 3, nt = kk & 3;
    if (4 * nt >= upw) return -1;
    return w * upw + 4 * nt + ul;
}

// A fragments of one 64-wide k block for every (wave, gate tile, digit): frag[((nt) * 3 + d) * 256 + tid]
// appended to `out` (16 bytes per thread).  row0 = first row of the block's source vector in M,
// src_upw / src_units describe that vector, out_upw / NT the kernel's gate tiling.
void q8_fragments(std::vector<int8_t>& out, const Q8Mat& M, int row0, int src_upw, int src_units, int H_out,
                  int out_upw, int NT)
{
    for (int nt = 0; nt < NT; nt++)
        for (int d = 0; d < 3; d++)
            for (int tid = 0; tid < LSTM_THREADS; tid++) {
                const int w = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
                const int col = (i & 3) * H_out + w * out_upw + nt * 4 + (i >> 2);
                for (int b = 0; b < 16; b++) {
                    const int unit = q8_unit_of_k(16 * g + b, src_upw);
                    int8_t v = 0;
                    if (unit >= 0 && unit < src_units) v = (int8_t)q8_digit(M.w[(size_t)(row0 + unit) * M.G + col], d);
                    out.push_back(v);
                }
            }
}

// One A fragment (16 x 64 weight digits of gate tile `nt` for every wave: 256 threads x 16 bytes) from a k map:
// `at(kpos)` says which row of M and which digit sits at k position `kpos` of the MFMA (row < 0: zero).  The gate
// rows are those of q8_fragments (H_out units, out_upw per wave).
template <typename F>
void q8_fragment_kmap(std::vector<int8_t>& out, const Q8Mat& M, int H_out, int out_upw, int nt, const F& at)
{
    for (int tid = 0; tid < LSTM_THREADS; tid++) {
        const int w = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
        const int col = (i & 3) * H_out + w * out_upw + nt * 4 + (i >> 2);
        for (int b = 0; b < 16; b++) {
            int row = -1, digit = 0;
            at(16 * g + b, row, digit);
            out.push_back(row < 0 ? (int8_t)0 : (int8_t)q8_digit(M.w[(size_t)row * M.G + col], digit));
        }
    }
}

// K2's ten fragments per gate tile (k_scaler_lstm_q8): the k positions of the shared hidden-vector row
//   block 0:  position 16 w + 4 ul + j, j < 3: layer-1 unit 12 w + 4 j + ul; j == 3: layer-2 unit 12 w + ul (nt 0)
//   block 1': position q = 2 (4 w + ul) + jj of its 32 bytes: layer-2 unit 12 w + 4 (1 + jj) + ul
// m1 = layer 1 (48 recurrent rows), m2 = layer 2 (rows 0-47 multiply h1, rows 48-95 its own h2).
void q8_scaler_fragments(std::vector<int8_t>& out, const Q8Mat& m1, const Q8Mat& m2)
{
    auto block0 = [](int kpos, int& l1_unit, int& l2_unit) {
        const int w = kpos >> 4, ulx = (kpos >> 2) & 3, j = kpos & 3;
        l1_unit = j < 3 ? 12 * w + 4 * j + ulx : -1;
        l2_unit = j == 3 ? 12 * w + ulx : -1;
    };
    auto half_unit = [](int q) { const int t = q >> 1, w = t >> 2, ulx = t & 3, jj = q & 1; return 12 * w + 4 * (1 + jj) + ulx; };
    for (int nt = 0; nt < 3; nt++) {
        for (int d = 2; d >= 0; d--)          // 0-2: layer 1, digits 2, 1, 0 over block 0 (zero at the layer-2 bytes)
            q8_fragment_kmap(out, m1, 48, 12, nt, [&](int k, int& row, int& digit) {
                int u1, u2; block0(k, u1, u2); row = u1; digit = d; });
        for (int d = 2; d >= 0; d--)          // 3-5: layer 2 over block 0: W2 rows at the layer-1 bytes, U2 rows at its own
            q8_fragment_kmap(out, m2, 48, 12, nt, [&](int k, int& row, int& digit) {
                int u1, u2; block0(k, u1, u2); row = u1 >= 0 ? u1 : 48 + u2; digit = d; });
        // 6-9: the half blocks of layer 2 (U2 rows): [first half digit | second half digit], -1 = zeros
        static const int halves[4][2] = { {2, 1}, {1, 0}, {-1, 2}, {-1, 0} };
        for (int f = 0; f < 4; f++)
            q8_fragment_kmap(out, m2, 48, 12, nt, [&](int k, int& row, int& digit) {
                const int dg = halves[f][k >> 5];
                row = dg < 0 ? -1 : 48 + half_unit(k & 31);
                digit = dg < 0 ? 0 : dg; });
    }
}

// The latency form's eight fragments per gate tile (k_lstm_q8_lat.hip): k position 16 g + b of block 0 = unit 16 g + b
// of h1 (g < 3) or unit b of h2 (g = 3); block 1' = h2 units 16-47 twice: its first two k groups under weight digit
// (2 | 0), its last two under digit (1 | none).
void q8_scaler_lat_fragments(std::vector<int8_t>& out, const Q8Mat& m1, const Q8Mat& m2)
{
    for (int nt = 0; nt < 3; nt++) {
        for (int d = 2; d >= 0; d--)          // 0-2: layer 1 over block 0 (zero at the h2 group)
            q8_fragment_kmap(out, m1, 48, 12, nt, [&](int k, int& row, int& digit) {
                row = k < 48 ? k : -1; digit = d; });
        for (int d = 2; d >= 0; d--)          // 3-5: layer 2 over block 0: W2 rows at h1, U2 rows 0-15 at the h2 group
            q8_fragment_kmap(out, m2, 48, 12, nt, [&](int k, int& row, int& digit) {
                row = k < 48 ? k : 48 + (k - 48); digit = d; });
        for (int f = 0; f < 2; f++)           // 6: [w2 | w2 | w1 | w1], 7: [w0 | w0 | 0 | 0] over h2 units 16-47
            q8_fragment_kmap(out, m2, 48, 12, nt, [&](int k, int& row, int& digit) {
                const int half = k >> 5;
                row = (f == 1 && half == 1) ? -1 : 48 + 16 + (k & 31);
                digit = f == 0 ? 2 - half : 0; });
    }
}

// K5a's latency form: [direction][gate tile][w2, w1, w0]: k position = unit (48 of 64 used)
void q8_bidir_lat_fragments(std::vector<int8_t>& out, const Q8Mat& mf, const Q8Mat& mb)
{
    for (int dir = 0; dir < 2; dir++)
        for (int nt = 0; nt < 3; nt++)
            for (int d = 2; d >= 0; d--)
                q8_fragment_kmap(out, dir ? mb : mf, 48, 12, nt, [&](int k, int& row, int& digit) {
                    row = k < 48 ? k : -1; digit = d; });
}

// K5b's latency form: [gate tile 0-3][8]: block 0 = hf 0-47 | hb 0-15, block 1 = hb 16-47 | h 0-31 (w2, w1, w0 each),
// then the half block h 32-63 as [w2 | w1] and [w0 | 0].  Rows of m3: 0-47 hf, 48-95 hb, 96-159 the cell's own h.
void q8_top_lat_fragments(std::vector<int8_t>& out, const Q8Mat& m3)
{
    for (int nt = 0; nt < 4; nt++) {
        for (int d = 2; d >= 0; d--)
            q8_fragment_kmap(out, m3, 64, 16, nt, [&](int k, int& row, int& digit) { row = k; digit = d; });
        for (int d = 2; d >= 0; d--)
            q8_fragment_kmap(out, m3, 64, 16, nt, [&](int k, int& row, int& digit) { row = 64 + k; digit = d; });
        for (int f = 0; f < 2; f++)
            q8_fragment_kmap(out, m3, 64, 16, nt, [&](int k, int& row, int& digit) {
                const int half = k >> 5;
                row = (f == 1 && half == 1) ? -1 : 128 + (k & 31);
                digit = f == 0 ? 2 - half : 0; });
    }
}

template <typename T>
int q8_to_device(pxg_ctx* ctx, T** dst, const std::vector<T>& src)
{
    if (*dst) (void)hipFree(*dst);
    *dst = nullptr;
    PXG_HIP(ctx, hipMalloc((void**)dst, src.size() * sizeof(T)));
    PXG_HIP(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return PXG_OK;
}

}  // namespace

// called from pxg_create while the config's host weight pointers are still valid
int pxg_q8_upload(pxg_ctx* ctx)
{
    const pxg_config& c = ctx->cfg;
    int rc;
    {   // scaler: U1 | W2 | U2
        const Q8Mat m1 = q8_quantise(c.scaler_lstm1), m2 = q8_quantise(c.scaler_lstm2);
        std::vector<int8_t> f;
        q8_scaler_fragments(f, m1, m2);
        if ((rc = q8_to_device(ctx, &ctx->q8.scaler_frag, f))) return rc;
        std::vector<int8_t> fl;
        q8_scaler_lat_fragments(fl, m1, m2);
        if ((rc = q8_to_device(ctx, &ctx->q8.scaler_frag_lat, fl))) return rc;
        m1.scales(ctx->q8.s_scaler1);
        m2.scales(ctx->q8.s_scaler2);
    }
    {   // demux bidirectional layer: forward U | backward U
        const Q8Mat mf = q8_quantise(c.demux_fwd), mb = q8_quantise(c.demux_bwd);
        std::vector<int8_t> f;
        q8_fragments(f, mf, 0, 12, 48, 48, 12, 3);
        q8_fragments(f, mb, 0, 12, 48, 48, 12, 3);
        if ((rc = q8_to_device(ctx, &ctx->q8.bidir_frag, f))) return rc;
        std::vector<int8_t> fl;
        q8_bidir_lat_fragments(fl, mf, mb);
        if ((rc = q8_to_device(ctx, &ctx->q8.bidir_frag_lat, fl))) return rc;
        mf.scales(ctx->q8.s_fwd);
        mb.scales(ctx->q8.s_bwd);
    }
    {   // demux top cell: rows 0-47 multiply hf, 48-95 hb, 96-159 its own h
        const Q8Mat m3 = q8_quantise(c.demux_top);
        std::vector<int8_t> f;
        q8_fragments(f, m3, 0, 12, 48, 64, 16, 4);
        q8_fragments(f, m3, 48, 12, 48, 64, 16, 4);
        q8_fragments(f, m3, 96, 16, 64, 64, 16, 4);
        if ((rc = q8_to_device(ctx, &ctx->q8.top_frag, f))) return rc;
        std::vector<int8_t> fl;
        q8_top_lat_fragments(fl, m3);
        if ((rc = q8_to_device(ctx, &ctx->q8.top_frag_lat, fl))) return rc;
        m3.scales(ctx->q8.s_top);
    }
    return PXG_OK;
}

void pxg_q8_free(pxg_ctx* ctx)
{
    if (ctx->q8.scaler_frag) (void)hipFree(ctx->q8.scaler_frag);
    if (ctx->q8.bidir_frag) (void)hipFree(ctx->q8.bidir_frag);
    if (ctx->q8.top_frag) (void)hipFree(ctx->q8.top_frag);
    if (ctx->q8.scaler_frag_lat) (void)hipFree(ctx->q8.scaler_frag_lat);
    if (ctx->q8.bidir_frag_lat) (void)hipFree(ctx->q8.bidir_frag_lat);
    if (ctx->q8.top_frag_lat) (void)hipFree(ctx->q8.top_frag_lat);
    ctx->q8.scaler_frag = ctx->q8.bidir_frag = ctx->q8.top_frag = nullptr;
    ctx->q8.scaler_frag_lat = ctx->q8.bidir_frag_lat = ctx->q8.top_frag_lat = nullptr;
    if (ctx->scaler_traj.p) (void)hipFree(ctx->scaler_traj.p);
    ctx->scaler_traj.p = nullptr;
    ctx->scaler_traj.cap = 0;
}

static int q8_scaler_launch(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx, const int32_t* count, const float* head,
                            float* pred, const int64_t* off, unsigned* traj_out);

// The zero-input trajectory of the scaler network, recorded by the kernel itself: one tile of 16 all-zero heads in
// step blocks of Q8S_TRAJ, every hand-over written to the table instead of the ring.  (T / Q8S_TRAJ + 1 entries of
// 12 KB: 6 MB for the shipped 2 000 steps; ~500 tasks on one workgroup, once per context.)
int pxg_q8_scaler_trajectory(pxg_ctx* ctx)
{
    if (!ctx->prefix_skip || ctx->scaler_traj.p) return PXG_OK;
    const int T = ctx->cfg.scaler_length / ctx->cfg.stride;
    if (T % 4) return PXG_OK;          // (the x tile is loaded as float4: the kernel's own requirement)
    DevBuf<float> zero_head, scratch_pred;
    int rc;
    if ((rc = pxg_reserve(ctx, zero_head, (size_t)16 * T)) || (rc = pxg_reserve(ctx, scratch_pred, 32)) ||
        (rc = pxg_reserve(ctx, ctx->scaler_traj, (size_t)(T / Q8S_TRAJ + 1) * Q8S_STATE)))
        return rc;
    PXG_HIP(ctx, hipMemsetAsync(zero_head.p, 0, (size_t)16 * T * sizeof(float), ctx->stream));
    PXG_HIP(ctx, hipMemsetAsync(ctx->scaler_traj.p, 0, (size_t)(T / Q8S_TRAJ + 1) * Q8S_STATE * sizeof(unsigned), ctx->stream));
    rc = q8_scaler_launch(ctx, 16, nullptr, nullptr, zero_head.p, scratch_pred.p, nullptr, ctx->scaler_traj.p);
    if (rc == PXG_OK) PXG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(zero_head.p);
    (void)hipFree(scratch_pred.p);
    return rc;
}

int pxg_launch_scaler_lstm_q8(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx, const int32_t* count,
                              const float* head, float* pred, const int64_t* off)
{
    // small batches: 4-read tiles on four times as many CUs (k_lstm_q8_lat.hip); PXG_K2_LAT_MAX = 0 switches it off
    const char* lat_env = getenv("PXG_K2_LAT_MAX");
    const int64_t lat_max = lat_env ? atoll(lat_env) : 8 * (int64_t)ctx->n_cu;
    if (n_rows <= lat_max) return pxg_launch_scaler_lstm_q8_lat(ctx, n_rows, idx, count, head, pred);
    return q8_scaler_launch(ctx, n_rows, idx, count, head, pred, off, nullptr);
}

static int q8_scaler_launch(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx, const int32_t* count, const float* head,
                            float* pred, const int64_t* off, unsigned* traj_out)
{
    if (n_rows <= 0) return PXG_OK;
    const int T = ctx->cfg.scaler_length / ctx->cfg.stride;
    const int64_t tiles = (n_rows + 15) / 16, slots = 2 * (int64_t)ctx->n_cu;
    const int64_t grid = std::min(tiles, slots);
    int rc;
    if (!ctx->lstm_err.p) {
        if ((rc = pxg_reserve(ctx, ctx->lstm_err, 1))) return rc;
        PXG_HIP(ctx, hipMemsetAsync(ctx->lstm_err.p, 0, sizeof(int), ctx->stream));
    }
    if ((rc = pxg_reserve(ctx, ctx->lstm_q, (size_t)(2 + tiles))) ||
        (rc = pxg_reserve(ctx, ctx->lstm_state, (size_t)2 * tiles * Q8S_STATE)))
        return rc;
    PXG_HIP(ctx, hipMemsetAsync(ctx->lstm_q.p, 0, (size_t)(2 + tiles) * sizeof(int), ctx->stream));
    const char* forced = getenv("PXG_SCALER_BLOCK_STEPS");           // tuning knob: steps per task of K2
    if (traj_out || forced) {
        ctx->q8.forced_block_steps = traj_out ? Q8S_TRAJ : atoi(forced);      // (the copy reads it later: a field of the context, not a local)
        PXG_HIP(ctx, hipMemcpyAsync(ctx->lstm_q.p + 1, &ctx->q8.forced_block_steps, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    }
    const unsigned* traj = (!traj_out && off && ctx->prefix_skip) ? reinterpret_cast<const unsigned*>(ctx->scaler_traj.p) : nullptr;
    const size_t lds = sizeof(float) * 4 * PXG_SIG_NSEG + 2 * Q2_HVEC + sizeof(float) * 16 * XS + 16 * 3 * 48 +
                       16 * 3 * Q2_LDS_FRAGS * LSTM_THREADS + sizeof(int) * 32;
    PXG_HIP(ctx, hipFuncSetAttribute((const void*)k_scaler_lstm_q8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const PxgLstmDev &l1 = ctx->scaler1, &l2 = ctx->scaler2;
    const Q8Scale s1 = q8_scale(ctx->q8.s_scaler1), s2 = q8_scale(ctx->q8.s_scaler2);
    hipLaunchKernelGGL(k_scaler_lstm_q8, dim3((unsigned)grid), dim3(LSTM_THREADS), lds, ctx->stream, (int)n_rows, idx,
                       count, T, head, ctx->d_sigtab, reinterpret_cast<const v4i*>(ctx->q8.scaler_frag), l1.kernel,
                       l1.bias, l2.bias, s1, s2, ctx->scaler_dense.kernel,
                       ctx->scaler_dense.bias, pred, ctx->lstm_q.p, ctx->lstm_err.p, ctx->lstm_q.p + 2,
                       reinterpret_cast<unsigned*>(ctx->lstm_state.p), off, ctx->cfg.scaler_length, ctx->cfg.stride, traj, traj_out);
    ctx->timeslice_used = true;
    PXG_HIP(ctx, hipGetLastError());
    return PXG_OK;
}
