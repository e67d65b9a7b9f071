// k_lstm.hip -- K2/K5: the two recurrent networks on fp32 MFMA (gfx950).
//   K2  scaler  LSTM(48,seq) -> LSTM(48) -> Dense(2)          a4  (signal_loader.py:96-97)
//   K5a demux   Bidirectional(LSTMCell 48) over 300 steps      a12 (barcoding.py:106-107)
//   K5b demux   LSTMCell(64) -> Dense(5) -> softmax            a12
//
// Why this shape (DESIGN.md "LSTM kernels"): per read the nets cost 148 MFLOP
// of strictly sequential small GEMMs, so the bound is fp32 matrix throughput
// and step latency, not HBM.  A 4-wave workgroup owns 16..64 reads (1-4 read
// tiles of 16) for the whole sequence.  The gate rows are split over the four
// waves so every wave keeps ITS weight slice in VGPRs for all steps (A operand
// of v_mfma_f32_16x16x4_f32: one VGPR per 16x4 block); hidden states are
// exchanged through LDS once per step (B operand, 3-4 ds_read_b128 per tile
// thanks to a k-major row layout); the c-state never leaves registers.  Layer
// 2 of the scaler runs one step behind layer 1 inside the same step so both
// matmuls share their B fragments.
//
// Z^T = W^T . [x|h]^T : the MFMA rows are gate rows and the columns are reads.
// A gate tile is 16 rows = 4 units x 4 gates (row = unit_local*4 + gate), and
// the C/D layout of the 16x16 MFMA gives lane l the rows 4*(l>>4)+{0..3} of
// column l&15 -- i.e. the FOUR GATES of unit (l>>4) for read (l&15) land in
// the four accumulator registers of one lane.  No cross-lane traffic at all:
// every lane activates i,f,g,o and updates one (read, unit) cell.
//
// Measured on MI355X (profiles/r01): the f32 MFMA shares the FP32 lanes with
// the VALU (busy cycles add up, they do not overlap), so the gate math is kept
// to ~60 VALU instructions per tile by a spline table in LDS (pxg_common.h).
//
// Canonical arithmetic (bit-exact with oracle/pxo_core.c lstm_step): the MFMA
// is a k-ordered fmaf chain, accumulator start = fl(fl(x*W)+b) (scalar input)
// or b, chain over input rows then recurrent rows; c' = fl(fl(f*c)+fl(i*g));
// h = fl(o*tanh(c')).
#include "pxg_common.h"
#include "k_lstm_shared.h"

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// k-major position of unit u inside a hidden-state row of H floats
template <int H>
__device__ __forceinline__ int hpos(int u)
{
    return (u & 3) * (H / 4) + (u >> 2);
}

// A fragments (weights) of one gate tile: rows [row0, row0+4*KB) of a Keras
// [rows, 4H] matrix; MFMA row i = lane&15 <-> column gate(i&3)*H + unit0 + (i>>2)
template <int H, int KB>
__device__ __forceinline__ void load_wfrag(float (&w)[KB], const float* __restrict__ mat,
                                           int row0, int unit0, int lane)
{
    const int k = lane >> 4, i = lane & 15;
    const int col = (i & 3) * H + unit0 + (i >> 2);
#pragma unroll
    for (int kb = 0; kb < KB; kb++) w[kb] = mat[(size_t)(row0 + kb * 4 + k) * (4 * H) + col];
}

// per-lane bias / scalar-input weight of the 4 gate rows this lane accumulates
template <int H>
__device__ __forceinline__ void load_gate4(float (&v)[4], const float* __restrict__ vec, int unit0,
                                           int lane)
{
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = vec[r * H + unit0 + (lane >> 4)];
}

// Hidden-state rows live in LDS with a stride of H+4 floats: 16 lanes reading 16
// consecutive rows with ds_read_b128 then touch 16 distinct bank quads (a stride
// of 48 or 64 floats maps them onto 4 resp. 1 quad: measured 64 % of the LDS
// cycles of K2 were bank conflicts, profiles/r01/e_*).
#define HSTRIDE(H) ((H) + 4)

// B fragments of one 16-read tile: hidden row (lane&15), k = lane>>4.
template <int H>
__device__ __forceinline__ void load_afrag(float (&a)[H / 4], const float* hrow_base, int lane)
{
    const float* p = hrow_base + (lane & 15) * HSTRIDE(H) + (lane >> 4) * (H / 4);
#pragma unroll
    for (int q = 0; q < H / 16; q++) {
        const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
        a[4 * q + 0] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
}

// Balanced static split of the read tiles over the launched workgroups: the
// first `rem` groups take one tile more.  With grid = 2 x #CU the dispatcher
// (measured: block b -> the CU of b mod #CU) pairs a heavy group with a light
// one per CU; placement only affects speed, never results.
__device__ __forceinline__ bool my_tiles(int lim_rows, int& tile0, int& ntile)
{
    const int n_tiles = (lim_rows + 15) >> 4;
    const int nb = min((int)gridDim.x, n_tiles);
    if ((int)blockIdx.x >= nb) return false;
    const int base = n_tiles / nb, rem = n_tiles % nb;
    const int b = blockIdx.x;
    ntile = base + (b < rem ? 1 : 0);
    tile0 = b * base + min(b, rem);
    return true;
}

// ===========================================================================
// K2: scaler.  wave = slice: units [12*slice, +12) of BOTH layers.
// ===========================================================================
template <int MTW>
__global__ __launch_bounds__(LSTM_THREADS, 2) void k_scaler_lstm(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const float* __restrict__ head, const float* __restrict__ sigtab,
    const float* __restrict__ W1, const float* __restrict__ U1, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ U2, const float* __restrict__ b2,
    const float* __restrict__ Wd, const float* __restrict__ bd, float* __restrict__ pred)
{
    constexpr int H = 48, NT = 3, KB = 12;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    int tile0, ntile;
    if (!my_tiles(lim, tile0, ntile)) return;
    const int row_base = tile0 * 16;

    float4* tab = reinterpret_cast<float4*>(smem);     // [1024] sigmoid spline
    constexpr int HS = HSTRIDE(H);
    float* h1 = smem + 4 * PXG_SIG_NSEG;               // [2][MTW][16][HS]
    float* h2 = h1 + 2 * MTW * 16 * HS;                // [2][MTW][16][HS]
    float* xb = h2 + 2 * MTW * 16 * HS;                // [16*MTW][XS]
    int* ridx = reinterpret_cast<int*>(xb + 16 * MTW * XS);   // [16*MTW]

    const int tid = threadIdx.x, lane = tid & 63, slice = tid >> 6;
    const int rd_l = lane & 15, ul = lane >> 4;        // read column, unit inside the tile

    load_sigtab(tab, sigtab, tid);
    for (int i = tid; i < 4 * MTW * 16 * HS; i += LSTM_THREADS) h1[i] = 0.0f;
    for (int i = tid; i < 16 * MTW; i += LSTM_THREADS) {
        const int row = row_base + i;
        ridx[i] = (i < 16 * ntile && row < lim) ? (idx ? idx[row] : row) : -1;
    }

    // ---- this wave's weight slice -> registers ------------------------------
    float wA[NT][KB], wB[NT][2 * KB], bias1[NT][4], bias2[NT][4], wx[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int unit0 = slice * 12 + nt * 4;
        load_wfrag<H, KB>(wA[nt], U1, 0, unit0, lane);
        float tmp[KB];
        load_wfrag<H, KB>(tmp, W2, 0, unit0, lane);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) wB[nt][kb] = tmp[kb];
        load_wfrag<H, KB>(tmp, U2, 0, unit0, lane);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) wB[nt][KB + kb] = tmp[kb];
        load_gate4<H>(bias1[nt], b1, unit0, lane);
        load_gate4<H>(bias2[nt], b2, unit0, lane);
        load_gate4<H>(wx[nt], W1, unit0, lane);
    }
    float c1[MTW][NT], c2[MTW][NT];
#pragma unroll
    for (int m = 0; m < MTW; m++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) c1[m][nt] = c2[m][nt] = 0.0f;
    __syncthreads();

    for (int t = 0; t <= T; t++) {
        if ((t % XCH) == 0 && t < T) {        // refill the x tile (rows x XCH steps)
            __syncthreads();
            for (int i = tid; i < 16 * MTW * (XCH / 4); i += LSTM_THREADS) {
                const int row = i / (XCH / 4), c4 = i % (XCH / 4);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int rd = ridx[row];
                if (rd >= 0 && t + c4 * 4 < T)
                    v = *reinterpret_cast<const float4*>(head + (size_t)rd * T + t + c4 * 4);
                *reinterpret_cast<float4*>(xb + row * XS + c4 * 4) = v;
            }
            __syncthreads();
        }
        const int rdb = t & 1, wrb = (t + 1) & 1;
#pragma unroll
        for (int m = 0; m < MTW; m++) {
            if (m < ntile) {
                float a1[KB], a2[KB];
                load_afrag<H>(a1, h1 + (rdb * MTW + m) * 16 * HS, lane);
                load_afrag<H>(a2, h2 + (rdb * MTW + m) * 16 * HS, lane);
                f32x4 acc1[NT], acc2[NT];
                const float x = xb[(m * 16 + rd_l) * XS + (t % XCH)];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float xw = x * wx[nt][r];
                        acc1[nt][r] = xw + bias1[nt][r];
                        acc2[nt][r] = bias2[nt][r];
                    }
                }
#pragma unroll
                for (int kb = 0; kb < KB; kb++) {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        acc1[nt] = mfma4(wA[nt][kb], a1[kb], acc1[nt]);
                        acc2[nt] = mfma4(wB[nt][kb], a1[kb], acc2[nt]);
                    }
                }
                float* o1 = h1 + ((wrb * MTW + m) * 16 + rd_l) * HS;
                float* o2 = h2 + ((wrb * MTW + m) * 16 + rd_l) * HS;
                if (t < T) {          // layer-1 activations under the last third of the MFMAs (see K2q)
                    float hn[NT];
                    cells_update<NT>(tab, acc1, c1[m], hn);
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) o1[hpos<H>(slice * 12 + nt * 4 + ul)] = hn[nt];
                }
#pragma unroll
                for (int kb = 0; kb < KB; kb++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                        acc2[nt] = mfma4(wB[nt][KB + kb], a2[kb], acc2[nt]);
                if (t >= 1) {
                    float hn[NT];
                    cells_update<NT>(tab, acc2, c2[m], hn);
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) o2[hpos<H>(slice * 12 + nt * 4 + ul)] = hn[nt];
                }
            }
        }
        __syncthreads();
    }
    // ---- Dense(2): chain over k = 0..47 from the bias ----------------------
    const int fin = (T + 1) & 1;
    for (int i = tid; i < 16 * ntile * 2; i += LSTM_THREADS) {
        const int row = i >> 1, j = i & 1;
        const int rd = ridx[row];
        if (rd < 0) continue;
        const float* hr = h2 + ((fin * MTW + (row >> 4)) * 16 + (row & 15)) * HS;
        float acc = bd[j];
        for (int k = 0; k < H; k++) acc = __builtin_fmaf(hr[hpos<H>(k)], Wd[k * 2 + j], acc);
        pred[(size_t)rd * 2 + j] = acc;
    }
}

// ===========================================================================
// K2q: the same scaler network, TIME-SLICED, used whenever there are more tiles than
// resident workgroups.  The static split leaves whole tiles of imbalance (10 000 reads:
// 625 tiles on 256 CUs -> 3 on some CUs, 2 on the others, 81 % busy; measured 13.3 ms
// static, 10.7 ms time-sliced).  Here every workgroup (2 per CU, all resident) pulls
// (step block, tile) tasks from a queue; a tile's LSTM state travels through HBM
// between blocks (12.8 KB), so the tiles advance round-robin on the 2 x #CU slots and
// the launch ends when the WORK runs out, not when the fullest CU does.  Tasks are
// ordered block-major, so the predecessor of a task was taken n_tiles tasks earlier;
// its completion is still checked through a per-tile counter (release/acquire at
// agent scope: the L2s of the 8 XCDs are not coherent with each other).  The step
// body is k_scaler_lstm<1>'s, results are bit-identical.
// ===========================================================================
#define QSTATE (2 * 16 * HSTRIDE(48) + 2 * LSTM_THREADS * 3)    // floats per saved tile state
__global__ __launch_bounds__(LSTM_THREADS, 2) void k_scaler_lstm_q(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const float* __restrict__ head, const float* __restrict__ sigtab,
    const float* __restrict__ W1, const float* __restrict__ U1, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ U2, const float* __restrict__ b2,
    const float* __restrict__ Wd, const float* __restrict__ bd, float* __restrict__ pred,
    int* __restrict__ queue /* [0] next task */, int* __restrict__ errflag, int* __restrict__ done /* per tile */,
    float* __restrict__ state /* [block][tile][QSTATE] */)
{
    constexpr int H = 48, NT = 3, KB = 12, HS = HSTRIDE(H);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    const int n_tiles = (lim + 15) >> 4;
    const int QBS = scaler_block_steps(n_tiles, (int)gridDim.x, T + 1);
    const int n_blocks = (T + 1 + QBS - 1) / QBS;
    const int n_tasks = n_tiles * n_blocks;

    float4* tab = reinterpret_cast<float4*>(smem);     // [1024] sigmoid spline
    float* h1 = smem + 4 * PXG_SIG_NSEG;               // [2][16][HS]
    float* h2 = h1 + 2 * 16 * HS;                      // [2][16][HS]
    float* xb = h2 + 2 * 16 * HS;                      // [16][XS]
    int* ridx = reinterpret_cast<int*>(xb + 16 * XS);  // [16]
    int* s_task = ridx + 16;

    const int tid = threadIdx.x, lane = tid & 63, slice = tid >> 6;
    const int rd_l = lane & 15, ul = lane >> 4;

    load_sigtab(tab, sigtab, tid);
    float wA[NT][KB], wB[NT][2 * KB], bias1[NT][4], bias2[NT][4], wx[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int unit0 = slice * 12 + nt * 4;
        load_wfrag<H, KB>(wA[nt], U1, 0, unit0, lane);
        float tmp[KB];
        load_wfrag<H, KB>(tmp, W2, 0, unit0, lane);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) wB[nt][kb] = tmp[kb];
        load_wfrag<H, KB>(tmp, U2, 0, unit0, lane);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) wB[nt][KB + kb] = tmp[kb];
        load_gate4<H>(bias1[nt], b1, unit0, lane);
        load_gate4<H>(bias2[nt], b2, unit0, lane);
        load_gate4<H>(wx[nt], W1, unit0, lane);
    }

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
            ridx[tid] = row < lim ? (idx ? idx[row] : row) : -1;
        }
        float c1[NT], c2[NT];
        // two alternating state slots per tile: block b + 2 overwrites block b's only after block
        // b + 1 -- which read it -- has published
        float* st_in = state + ((size_t)((blk - 1) & 1) * n_tiles + tile) * QSTATE;
        if (blk == 0) {
            for (int i = tid; i < 4 * 16 * HS; i += LSTM_THREADS) h1[i] = 0.0f;    // h1 and h2, both buffers
#pragma unroll
            for (int nt = 0; nt < NT; nt++) c1[nt] = c2[nt] = 0.0f;
        } else {
            if (tid == 0) {                    // predecessor block of this tile published?
                int spins = 0;
                while (__hip_atomic_load(&done[tile], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < blk) {
                    __builtin_amdgcn_s_sleep(32);
                    if (++spins > (1 << 24)) { atomicExch(errflag, 1); break; }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int rb = t0 & 1;
            for (int i = tid; i < 16 * HS; i += LSTM_THREADS) {
                h1[rb * 16 * HS + i] = st_in[i];
                h2[rb * 16 * HS + i] = st_in[16 * HS + i];
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                c1[nt] = st_in[2 * 16 * HS + nt * LSTM_THREADS + tid];
                c2[nt] = st_in[2 * 16 * HS + (NT + nt) * LSTM_THREADS + tid];
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
            const int rdb = t & 1, wrb = (t + 1) & 1;
            float a1[KB], a2[KB];
            load_afrag<H>(a1, h1 + rdb * 16 * HS, lane);
            load_afrag<H>(a2, h2 + rdb * 16 * HS, lane);
            f32x4 acc1[NT], acc2[NT];
            const float x = xb[rd_l * XS + ((t - t0) % XCH)];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float xw = x * wx[nt][r];
                    acc1[nt][r] = xw + bias1[nt][r];
                    acc2[nt][r] = bias2[nt][r];
                }
            }
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    acc1[nt] = mfma4(wA[nt][kb], a1[kb], acc1[nt]);
                    acc2[nt] = mfma4(wB[nt][kb], a1[kb], acc2[nt]);
                }
            }
            // layer 1 is complete here: its activations (LDS table rows) are requested before the
            // last third of the matrix work, so their round trips run under those 36 MFMAs
            // (measured: 10.49 -> 10.30 ms; the same split of K5b's four gate tiles into two halves
            // leaves only two independent accumulator chains per phase and is SLOWER, 2.28 -> 2.41)
            float* o1 = h1 + (wrb * 16 + rd_l) * HS;
            float* o2 = h2 + (wrb * 16 + rd_l) * HS;
            if (t < T) {
                float hn[NT];
                cells_update<NT>(tab, acc1, c1, hn);
#pragma unroll
                for (int nt = 0; nt < NT; nt++) o1[hpos<H>(slice * 12 + nt * 4 + ul)] = hn[nt];
            }
#pragma unroll
            for (int kb = 0; kb < KB; kb++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    acc2[nt] = mfma4(wB[nt][KB + kb], a2[kb], acc2[nt]);
            if (t >= 1) {
                float hn[NT];
                cells_update<NT>(tab, acc2, c2, hn);
#pragma unroll
                for (int nt = 0; nt < NT; nt++) o2[hpos<H>(slice * 12 + nt * 4 + ul)] = hn[nt];
            }
            __syncthreads();
        }

        if (t1 == T + 1) {
            // ---- Dense(2): chain over k = 0..47 from the bias ----------------------
            const int fin = (T + 1) & 1;
            for (int i = tid; i < 16 * 2; i += LSTM_THREADS) {
                const int row = i >> 1, j = i & 1;
                const int rd = ridx[row];
                if (rd < 0) continue;
                const float* hr = h2 + (fin * 16 + row) * HS;
                float acc = bd[j];
                for (int k = 0; k < H; k++) acc = __builtin_fmaf(hr[hpos<H>(k)], Wd[k * 2 + j], acc);
                pred[(size_t)rd * 2 + j] = acc;
            }
        } else {
            // ---- hand the tile over: state of iteration t1 -> HBM, then publish -------
            float* st_out = state + ((size_t)(blk & 1) * n_tiles + tile) * QSTATE;
            const int rb = t1 & 1;
            for (int i = tid; i < 16 * HS; i += LSTM_THREADS) {
                st_out[i] = h1[rb * 16 * HS + i];
                st_out[16 * HS + i] = h2[rb * 16 * HS + i];
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                st_out[2 * 16 * HS + nt * LSTM_THREADS + tid] = c1[nt];
                st_out[2 * 16 * HS + (NT + nt) * LSTM_THREADS + tid] = c2[nt];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&done[tile], blk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ===========================================================================
// K5a: demux bidirectional layer.  The two "cells" are the forward net at step
// t and the backward net at step T-1-t.  Every step's hidden rows are streamed
// to HBM (k-major layout) for K5b.
// ===========================================================================
template <int MTW>
__global__ __launch_bounds__(LSTM_THREADS, 2) void k_demux_bidir(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const float* __restrict__ win, const float* __restrict__ sigtab,
    const float* __restrict__ Wf, const float* __restrict__ Uf, const float* __restrict__ bf,
    const float* __restrict__ Wb, const float* __restrict__ Ub, const float* __restrict__ bb,
    float* __restrict__ bidir)
{
    constexpr int H = 48, NT = 3, KB = 12;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    int tile0, ntile;
    if (!my_tiles(lim, tile0, ntile)) return;
    const int row_base = tile0 * 16;

    float4* tab = reinterpret_cast<float4*>(smem);
    constexpr int HS = HSTRIDE(H);
    float* hf = smem + 4 * PXG_SIG_NSEG;               // [2][MTW][16][HS]
    float* hb = hf + 2 * MTW * 16 * HS;
    float* xf = hb + 2 * MTW * 16 * HS;                // [16*MTW][XS]  x[t0 + c]
    float* xr = xf + 16 * MTW * XS;                    // [16*MTW][XS]  x[T-1-(t0+c)]
    int* ridx = reinterpret_cast<int*>(xr + 16 * MTW * XS);

    const int tid = threadIdx.x, lane = tid & 63, slice = tid >> 6;
    const int rd_l = lane & 15, ul = lane >> 4;

    load_sigtab(tab, sigtab, tid);
    for (int i = tid; i < 4 * MTW * 16 * HS; i += LSTM_THREADS) hf[i] = 0.0f;
    for (int i = tid; i < 16 * MTW; i += LSTM_THREADS) {
        const int row = row_base + i;
        ridx[i] = (i < 16 * ntile && row < lim) ? (idx ? idx[row] : row) : -1;
    }

    float wF[NT][KB], wBk[NT][KB], biasf[NT][4], biasb[NT][4], wxf[NT][4], wxb[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int unit0 = slice * 12 + nt * 4;
        load_wfrag<H, KB>(wF[nt], Uf, 0, unit0, lane);
        load_wfrag<H, KB>(wBk[nt], Ub, 0, unit0, lane);
        load_gate4<H>(biasf[nt], bf, unit0, lane);
        load_gate4<H>(biasb[nt], bb, unit0, lane);
        load_gate4<H>(wxf[nt], Wf, unit0, lane);
        load_gate4<H>(wxb[nt], Wb, unit0, lane);
    }
    float cf[MTW][NT], cb[MTW][NT];
#pragma unroll
    for (int m = 0; m < MTW; m++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) cf[m][nt] = cb[m][nt] = 0.0f;
    __syncthreads();

    // streaming of the hidden rows to HBM: the (row, direction, float4) a thread moves every
    // step is fixed, so the index arithmetic is done once (it was ~20 % of this kernel's VALU work)
    constexpr int NST = (MTW * 16 * 2 * (H / 4) + LSTM_THREADS - 1) / LSTM_THREADS;
    int st_src[NST];              // LDS float offset inside one (buffer, direction) plane, -1: idle
    int st_dir[NST];
    float* st_dst[NST];           // bidir + (rd * T) * 2H + dir * H + c4 * 4
#pragma unroll
    for (int p = 0; p < NST; p++) {
        const int i = tid + p * LSTM_THREADS;
        st_src[p] = -1; st_dir[p] = 0; st_dst[p] = bidir;
        if (i < 16 * ntile * 2 * (H / 4)) {
            const int c4 = i % (H / 4);
            const int dir = (i / (H / 4)) & 1;
            const int row = i / (2 * (H / 4));
            const int rd = ridx[row];
            if (rd >= 0) {
                st_src[p] = ((row >> 4) * 16 + (row & 15)) * HS + c4 * 4;
                st_dir[p] = dir;
                st_dst[p] = bidir + ((size_t)rd * T) * (2 * H) + dir * H + c4 * 4;
            }
        }
    }

    for (int t = 0; t <= T; t++) {
        const int rdb = t & 1, wrb = (t + 1) & 1;
        if (t >= 1) {    // stream the rows written in the previous step to HBM
            const int tf = t - 1, tb = T - t;
#pragma unroll
            for (int p = 0; p < NST; p++) {
                if (st_src[p] >= 0) {
                    const float* src = (st_dir[p] ? hb : hf) + rdb * MTW * 16 * HS + st_src[p];
                    float* dst = st_dst[p] + (size_t)(st_dir[p] ? tb : tf) * (2 * H);
                    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
                }
            }
        }
        if (t == T) break;
        if ((t % XCH) == 0) {                 // refill both x tiles
            __syncthreads();
            for (int i = tid; i < 16 * ntile * XCH; i += LSTM_THREADS) {
                const int row = i / XCH, c = i % XCH;
                const int rd = ridx[row];
                const int tt = t + c;
                const bool ok = rd >= 0 && tt < T;
                xf[row * XS + c] = ok ? win[(size_t)rd * T + tt] : 0.0f;
                xr[row * XS + c] = ok ? win[(size_t)rd * T + (T - 1 - tt)] : 0.0f;
            }
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < MTW; m++) {
            if (m < ntile) {
                float a1[KB], a2[KB];
                load_afrag<H>(a1, hf + (rdb * MTW + m) * 16 * HS, lane);
                load_afrag<H>(a2, hb + (rdb * MTW + m) * 16 * HS, lane);
                f32x4 acc1[NT], acc2[NT];
                const float x1 = xf[(m * 16 + rd_l) * XS + (t % XCH)];
                const float x2 = xr[(m * 16 + rd_l) * XS + (t % XCH)];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float p1 = x1 * wxf[nt][r];
                        acc1[nt][r] = p1 + biasf[nt][r];
                        const float p2 = x2 * wxb[nt][r];
                        acc2[nt][r] = p2 + biasb[nt][r];
                    }
                }
#pragma unroll
                for (int kb = 0; kb < KB; kb++) {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        acc1[nt] = mfma4(wF[nt][kb], a1[kb], acc1[nt]);
                        acc2[nt] = mfma4(wBk[nt][kb], a2[kb], acc2[nt]);
                    }
                }
                const int ob = ((wrb * MTW + m) * 16 + rd_l) * HS;
                float hn1[NT], hn2[NT];
                cells_update<NT>(tab, acc1, cf[m], hn1);
                cells_update<NT>(tab, acc2, cb[m], hn2);
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const int hp = ob + hpos<H>(slice * 12 + nt * 4 + ul);
                    hf[hp] = hn1[nt];
                    hb[hp] = hn2[nt];
                }
            }
        }
        __syncthreads();
    }
}

// ===========================================================================
// K5b: demux top cell (H=64, input 96) + Dense(5) + softmax.  wave = slice of
// 16 units (4 gate tiles, K = 96 input rows + 64 recurrent rows).
// ===========================================================================
template <int MTW>
__global__ __launch_bounds__(LSTM_THREADS, 2) void k_demux_top(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const float* __restrict__ bidir, const float* __restrict__ sigtab,
    const float* __restrict__ W3, const float* __restrict__ U3, const float* __restrict__ b3,
    const float* __restrict__ Wd, const float* __restrict__ bd, int n_classes,
    float* __restrict__ probs)
{
    constexpr int H = 64, HI = 48, NT = 4, KBI = 24, KBR = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    int tile0, ntile;
    if (!my_tiles(lim, tile0, ntile)) return;
    const int row_base = tile0 * 16;

    float4* tab = reinterpret_cast<float4*>(smem);
    constexpr int HS = HSTRIDE(H), IS = HSTRIDE(2 * HI);
    float* h3 = smem + 4 * PXG_SIG_NSEG;               // [2][MTW][16][HS]
    float* inb = h3 + 2 * MTW * 16 * HS;               // [2][MTW*16][IS]
    float* bl = inb + 2 * MTW * 16 * IS;               // [H][4] bias, (i,f,g,o) per unit
    int* ridx = reinterpret_cast<int*>(bl + 4 * H);

    const int tid = threadIdx.x, lane = tid & 63, slice = tid >> 6;
    const int rd_l = lane & 15, ul = lane >> 4;

    load_sigtab(tab, sigtab, tid);
    for (int i = tid; i < 2 * MTW * 16 * HS; i += LSTM_THREADS) h3[i] = 0.0f;
    for (int i = tid; i < 4 * H; i += LSTM_THREADS) bl[i] = b3[(i & 3) * H + (i >> 2)];
    for (int i = tid; i < 16 * MTW; i += LSTM_THREADS) {
        const int row = row_base + i;
        ridx[i] = (i < 16 * ntile && row < lim) ? (idx ? idx[row] : row) : -1;
    }
    float wI[NT][KBI], wR[NT][KBR];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int unit0 = slice * 16 + nt * 4;
        load_wfrag<H, KBI>(wI[nt], W3, 0, unit0, lane);
        load_wfrag<H, KBR>(wR[nt], U3, 0, unit0, lane);
    }
    float c3[MTW][NT];
#pragma unroll
    for (int m = 0; m < MTW; m++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) c3[m][nt] = 0.0f;
    __syncthreads();

    const int n_f4 = 16 * ntile * (2 * HI / 4);     // float4 per step of input rows
    for (int i = tid; i < n_f4; i += LSTM_THREADS) {
        const int row = i / (2 * HI / 4), c4 = i % (2 * HI / 4);
        const int rd = ridx[row];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rd >= 0) v = *reinterpret_cast<const float4*>(bidir + ((size_t)rd * T) * (2 * HI) + c4 * 4);
        *reinterpret_cast<float4*>(inb + row * IS + c4 * 4) = v;
    }
    __syncthreads();

    // per-thread constants of the input prefetch (fixed (row, float4) per thread: no index
    // arithmetic inside the step loop)
    constexpr int NPF = (MTW * 16 * (2 * HI / 4) + LSTM_THREADS - 1) / LSTM_THREADS;
    const float* pf_src[NPF];     // bidir + (rd * T) * 2HI + c4 * 4, nullptr: idle / padding row
    int pf_dst[NPF];              // LDS float offset inside one input buffer, -1: idle
#pragma unroll
    for (int p = 0; p < NPF; p++) {
        const int i = tid + p * LSTM_THREADS;
        pf_src[p] = nullptr;
        pf_dst[p] = -1;
        if (i < n_f4) {
            const int row = i / (2 * HI / 4), c4 = i % (2 * HI / 4);
            const int rd = ridx[row];
            pf_dst[p] = row * IS + c4 * 4;
            if (rd >= 0) pf_src[p] = bidir + ((size_t)rd * T) * (2 * HI) + c4 * 4;
        }
    }

    for (int t = 0; t < T; t++) {
        const int rdb = t & 1, wrb = (t + 1) & 1;
        // next step's input rows: issue the global loads now, park them in
        // registers while the MFMAs run, write them to the other LDS buffer
        // just before the barrier
        float4 pf[NPF];
#pragma unroll
        for (int p = 0; p < NPF; p++) {
            pf[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pf_src[p] != nullptr && t + 1 < T)
                pf[p] = *reinterpret_cast<const float4*>(pf_src[p] + (size_t)(t + 1) * (2 * HI));
        }
#pragma unroll
        for (int m = 0; m < MTW; m++) {
            if (m < ntile) {
                f32x4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const float4 bv = *reinterpret_cast<const float4*>(bl + (slice * 16 + nt * 4 + ul) * 4);
                    acc[nt][0] = bv.x; acc[nt][1] = bv.y; acc[nt][2] = bv.z; acc[nt][3] = bv.w;
                }
                const float* p = inb + (rdb * MTW * 16 + m * 16) * IS + rd_l * IS +
                                 (lane >> 4) * (HI / 4);
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    float a[HI / 4];
#pragma unroll
                    for (int v4 = 0; v4 < HI / 16; v4++) {
                        const float4 v = *reinterpret_cast<const float4*>(p + half * HI + 4 * v4);
                        a[4 * v4] = v.x; a[4 * v4 + 1] = v.y; a[4 * v4 + 2] = v.z; a[4 * v4 + 3] = v.w;
                    }
#pragma unroll
                    for (int kb = 0; kb < HI / 4; kb++)
#pragma unroll
                        for (int nt = 0; nt < NT; nt++)
                            acc[nt] = mfma4(wI[nt][half * (HI / 4) + kb], a[kb], acc[nt]);
                }
                {
                    float a[KBR];
                    load_afrag<H>(a, h3 + (rdb * MTW + m) * 16 * HS, lane);
#pragma unroll
                    for (int kb = 0; kb < KBR; kb++)
#pragma unroll
                        for (int nt = 0; nt < NT; nt++) acc[nt] = mfma4(wR[nt][kb], a[kb], acc[nt]);
                }
                float* o3 = h3 + ((wrb * MTW + m) * 16 + rd_l) * HS;
                float hn[NT];
#pragma unroll
                for (int half = 0; half < 2; half++) {      // 2 x 2 cells: 160 weight VGPRs leave no
                    f32x4 a2[2] = { acc[2 * half], acc[2 * half + 1] };   // room for 16 rows in flight
                    float cc[2] = { c3[m][2 * half], c3[m][2 * half + 1] };
                    float hh[2];
                    cells_update<2>(tab, a2, cc, hh);
                    c3[m][2 * half] = cc[0]; c3[m][2 * half + 1] = cc[1];
                    hn[2 * half] = hh[0]; hn[2 * half + 1] = hh[1];
                }
#pragma unroll
                for (int nt = 0; nt < NT; nt++) o3[hpos<H>(slice * 16 + nt * 4 + ul)] = hn[nt];
            }
        }
#pragma unroll
        for (int p = 0; p < NPF; p++)
            if (pf_dst[p] >= 0)
                *reinterpret_cast<float4*>(inb + wrb * MTW * 16 * IS + pf_dst[p]) = pf[p];
        __syncthreads();
    }
    // ---- Dense(n_classes) + softmax ----------------------------------------
    const int fin = T & 1;
    for (int row = tid; row < 16 * ntile; row += LSTM_THREADS) {
        const int rd = ridx[row];
        if (rd < 0) continue;
        const float* hr = h3 + ((fin * MTW + (row >> 4)) * 16 + (row & 15)) * HS;
        float z[PXG_MAX_CLASSES], e[PXG_MAX_CLASSES];
#pragma unroll
        for (int j = 0; j < PXG_MAX_CLASSES; j++) {
            z[j] = -__builtin_inff();
            if (j < n_classes) {
                float acc = bd[j];
                for (int k = 0; k < H; k++) acc = __builtin_fmaf(hr[hpos<H>(k)], Wd[k * n_classes + j], acc);
                z[j] = acc;
            }
        }
        float mx = z[0];
#pragma unroll
        for (int j = 1; j < PXG_MAX_CLASSES; j++) mx = (j < n_classes && z[j] > mx) ? z[j] : mx;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < PXG_MAX_CLASSES; j++) {
            e[j] = j < n_classes ? pxg_expf(z[j] - mx) : 0.0f;
            if (j < n_classes) s = (j == 0) ? e[0] : s + e[j];
        }
#pragma unroll
        for (int j = 0; j < PXG_MAX_CLASSES; j++)
            probs[(size_t)rd * PXG_MAX_CLASSES + j] = j < n_classes ? e[j] / s : 0.0f;
    }
}

// ===========================================================================
// K5a-q / K5b-q: the demux kernels TIME-SLICED like K2q, used whenever the batch has more
// 16-read tiles than resident workgroups.  With the static split a 10 000-read batch puts 2
// tiles on 107 of the 512 workgroups and 1 on the rest, and every step of a 2-tile group
// costs two tile-steps: the launch lasts 2 x 300 tile-steps for 619 x 300 / 512 = 363 of
// work per slot (60 %).  Here the 2 x #CU resident workgroups pull (step block, tile) tasks
// from a queue; the number of blocks per tile is chosen ON THE DEVICE from the actual tile
// count (demux_blocks below).  A tile's state (hidden rows + cell registers) travels
// through HBM between its blocks, two alternating slots per tile; hand-over as in K2q
// (per-tile progress counter, release / acquire at agent scope).  Weights stay in VGPRs
// across tasks.  The step bodies are those of k_demux_bidir<1> / k_demux_top<1>:
// bit-identical results.
// ===========================================================================
#define DQA_STATE (2 * 16 * HSTRIDE(48) + 2 * LSTM_THREADS * 3)     // hf, hb rows + cf, cb registers

// MT = read tiles per task.  Only MT = 1 is launched: a task chain is one tile group, so
// pairs halve the number of independent chains (619 tiles -> 310 groups for 512 slots) and
// measured slower at every batch size tried (10 000 reads: 2.33 + 2.97 ms against 1.56 + 2.29;
// 100 000 reads: 13.4 + 19.6 against 13.1 + 19.7).  An 8-wave variant of K5a (forward cell on
// waves 0-3, backward on 4-7, 128 VGPRs, four waves per SIMD) measured the same 1.57 ms as
// this one: the kernel is not latency-bound, see DESIGN.md section 3.1.
template <int MT>
__global__ __launch_bounds__(LSTM_THREADS, 2) void k_demux_bidir_q(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const float* __restrict__ win, const float* __restrict__ sigtab,
    const float* __restrict__ Wf, const float* __restrict__ Uf, const float* __restrict__ bf,
    const float* __restrict__ Wb, const float* __restrict__ Ub, const float* __restrict__ bb,
    float* __restrict__ bidir, int* __restrict__ queue, int* __restrict__ errflag,
    int* __restrict__ done /* per tile group */, float* __restrict__ state /* [2][tile][DQA_STATE] */)
{
    constexpr int H = 48, NT = 3, KB = 12, HS = HSTRIDE(H);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    const int n_tiles = (lim + 15) >> 4;
    const int n_groups = (n_tiles + MT - 1) / MT;
    const int n_blocks = queue[1] > 0 ? min(queue[1], T) : demux_blocks(n_groups, (int)gridDim.x, T);
    const int QB = (T + n_blocks - 1) / n_blocks;
    const int n_tasks = n_groups * n_blocks;

    float4* tab = reinterpret_cast<float4*>(smem);
    float* hf = smem + 4 * PXG_SIG_NSEG;               // [2][MT][16][HS]
    float* hb = hf + 2 * MT * 16 * HS;
    float* xf = hb + 2 * MT * 16 * HS;                 // [16*MT][XS]  x[c0 + c]
    float* xr = xf + 16 * MT * XS;                     // [16*MT][XS]  x[T-1-(c0+c)]
    int* ridx = reinterpret_cast<int*>(xr + 16 * MT * XS);  // [16*MT]
    int* s_task = ridx + 16 * MT;

    const int tid = threadIdx.x, lane = tid & 63, slice = tid >> 6;
    const int rd_l = lane & 15, ul = lane >> 4;

    load_sigtab(tab, sigtab, tid);
    float wF[NT][KB], wBk[NT][KB], biasf[NT][4], biasb[NT][4], wxf[NT][4], wxb[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int unit0 = slice * 12 + nt * 4;
        load_wfrag<H, KB>(wF[nt], Uf, 0, unit0, lane);
        load_wfrag<H, KB>(wBk[nt], Ub, 0, unit0, lane);
        load_gate4<H>(biasf[nt], bf, unit0, lane);
        load_gate4<H>(biasb[nt], bb, unit0, lane);
        load_gate4<H>(wxf[nt], Wf, unit0, lane);
        load_gate4<H>(wxb[nt], Wb, unit0, lane);
    }
    constexpr int NST = (MT * 16 * 2 * (H / 4) + LSTM_THREADS - 1) / LSTM_THREADS;

    for (;;) {
        __syncthreads();                       // everybody is done with the previous task's LDS
        if (tid == 0) *s_task = atomicAdd(&queue[0], 1);
        __syncthreads();
        const int q = *s_task;
        if (q >= n_tasks) break;
        const int blk = q / n_groups, grp = q % n_groups;
        const int tile0 = grp * MT;
        const int ntile = min(MT, n_tiles - tile0);
        const int t0 = blk * QB;
        const int t1 = min(t0 + QB, T);
        for (int i = tid; i < 16 * MT; i += LSTM_THREADS) {
            const int row = tile0 * 16 + i;
            ridx[i] = (i < 16 * ntile && row < lim) ? (idx ? idx[row] : row) : -1;
        }
        float cf[MT][NT], cb[MT][NT];
        if (blk == 0) {
            for (int i = tid; i < 4 * MT * 16 * HS; i += LSTM_THREADS) hf[i] = 0.0f;     // hf and hb, both buffers
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) cf[m][nt] = cb[m][nt] = 0.0f;
            __syncthreads();
        } else {
            dq_wait(done, grp, blk, errflag, tid);
            const int rb = t0 & 1;
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const float* st_in = state + ((size_t)((blk - 1) & 1) * n_tiles + min(tile0 + m, n_tiles - 1)) * DQA_STATE;
                for (int i = tid; i < 16 * HS; i += LSTM_THREADS) {
                    hf[(rb * MT + m) * 16 * HS + i] = st_in[i];
                    hb[(rb * MT + m) * 16 * HS + i] = st_in[16 * HS + i];
                }
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    cf[m][nt] = st_in[2 * 16 * HS + nt * LSTM_THREADS + tid];
                    cb[m][nt] = st_in[2 * 16 * HS + (NT + nt) * LSTM_THREADS + tid];
                }
            }
            __syncthreads();
        }
        // streaming of the hidden rows to HBM: fixed (row, direction, float4) per thread and task
        int st_src[NST], st_dir[NST];
        float* st_dst[NST];
#pragma unroll
        for (int p = 0; p < NST; p++) {
            const int i = tid + p * LSTM_THREADS;
            st_src[p] = -1; st_dir[p] = 0; st_dst[p] = bidir;
            if (i < 16 * ntile * 2 * (H / 4)) {
                const int c4 = i % (H / 4);
                const int dir = (i / (H / 4)) & 1;
                const int row = i / (2 * (H / 4));
                const int rd = ridx[row];
                if (rd >= 0) {
                    st_src[p] = row * HS + c4 * 4;
                    st_dir[p] = dir;
                    st_dst[p] = bidir + ((size_t)rd * T) * (2 * H) + dir * H + c4 * 4;
                }
            }
        }

        for (int t = t0; t <= t1; t++) {
            const int rdb = t & 1, wrb = (t + 1) & 1;
            if (t > t0) {    // stream the rows written in the previous step to HBM
                const int tf = t - 1, tb = T - t;
#pragma unroll
                for (int p = 0; p < NST; p++) {
                    if (st_src[p] >= 0) {
                        const float* src = (st_dir[p] ? hb : hf) + rdb * MT * 16 * HS + st_src[p];
                        float* dst = st_dst[p] + (size_t)(st_dir[p] ? tb : tf) * (2 * H);
                        {   // streamed once, read by the next kernel: keep the rows out of the L2's dirty set
                            const f32x4 v = *reinterpret_cast<const f32x4*>(src);
                            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
                        }
                    }
                }
            }
            if (t == t1) break;
            if (((t - t0) % XCH) == 0) {          // refill both x tiles
                __syncthreads();
                for (int i = tid; i < 16 * ntile * XCH; i += LSTM_THREADS) {
                    const int row = i / XCH, c = i % XCH;
                    const int rd = ridx[row];
                    const int tt = t + c;
                    const bool ok = rd >= 0 && tt < T;
                    xf[row * XS + c] = ok ? win[(size_t)rd * T + tt] : 0.0f;
                    xr[row * XS + c] = ok ? win[(size_t)rd * T + (T - 1 - tt)] : 0.0f;
                }
                __syncthreads();
            }
#pragma unroll
            for (int m = 0; m < MT; m++) {
                if (m < ntile) {
                    float a1[KB], a2[KB];
                    load_afrag<H>(a1, hf + (rdb * MT + m) * 16 * HS, lane);
                    load_afrag<H>(a2, hb + (rdb * MT + m) * 16 * HS, lane);
                    f32x4 acc1[NT], acc2[NT];
                    const float x1 = xf[(m * 16 + rd_l) * XS + ((t - t0) % XCH)];
                    const float x2 = xr[(m * 16 + rd_l) * XS + ((t - t0) % XCH)];
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float p1 = x1 * wxf[nt][r];
                            acc1[nt][r] = p1 + biasf[nt][r];
                            const float p2 = x2 * wxb[nt][r];
                            acc2[nt][r] = p2 + biasb[nt][r];
                        }
                    }
#pragma unroll
                    for (int kb = 0; kb < KB; kb++) {
#pragma unroll
                        for (int nt = 0; nt < NT; nt++) {
                            acc1[nt] = mfma4(wF[nt][kb], a1[kb], acc1[nt]);
                            acc2[nt] = mfma4(wBk[nt][kb], a2[kb], acc2[nt]);
                        }
                    }
                    const int ob = ((wrb * MT + m) * 16 + rd_l) * HS;
                    float hn1[NT], hn2[NT];
                    cells_update<NT>(tab, acc1, cf[m], hn1);
                    cells_update<NT>(tab, acc2, cb[m], hn2);
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        const int hp = ob + hpos<H>(slice * 12 + nt * 4 + ul);
                        hf[hp] = hn1[nt];
                        hb[hp] = hn2[nt];
                    }
                }
            }
            __syncthreads();
        }

        if (t1 < T) {        // hand the tiles over: state of iteration t1 -> HBM, then publish
            const int rb = t1 & 1;
#pragma unroll
            for (int m = 0; m < MT; m++) {
                if (m < ntile) {
                    float* st_out = state + ((size_t)(blk & 1) * n_tiles + tile0 + m) * DQA_STATE;
                    for (int i = tid; i < 16 * HS; i += LSTM_THREADS) {
                        st_out[i] = hf[(rb * MT + m) * 16 * HS + i];
                        st_out[16 * HS + i] = hb[(rb * MT + m) * 16 * HS + i];
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        st_out[2 * 16 * HS + nt * LSTM_THREADS + tid] = cf[m][nt];
                        st_out[2 * 16 * HS + (NT + nt) * LSTM_THREADS + tid] = cb[m][nt];
                    }
                }
            }
            dq_publish(done, grp, blk, tid);
        }
    }
}

#define DQB_STATE (16 * HSTRIDE(64) + LSTM_THREADS * 4)              // h3 rows + c3 registers

template <int MT>
__global__ __launch_bounds__(LSTM_THREADS, 2) void k_demux_top_q(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const float* __restrict__ bidir, const float* __restrict__ sigtab,
    const float* __restrict__ W3, const float* __restrict__ U3, const float* __restrict__ b3,
    const float* __restrict__ Wd, const float* __restrict__ bd, int n_classes,
    float* __restrict__ probs, int* __restrict__ queue, int* __restrict__ errflag,
    int* __restrict__ done, float* __restrict__ state /* [2][tile][DQB_STATE] */)
{
    constexpr int H = 64, HI = 48, NT = 4, KBI = 24, KBR = 16;
    constexpr int HS = HSTRIDE(H), IS = HSTRIDE(2 * HI);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    const int n_tiles = (lim + 15) >> 4;
    const int n_groups = (n_tiles + MT - 1) / MT;
    const int n_blocks = queue[1] > 0 ? min(queue[1], T) : demux_blocks(n_groups, (int)gridDim.x, T);
    const int QB = (T + n_blocks - 1) / n_blocks;
    const int n_tasks = n_groups * n_blocks;

    float4* tab = reinterpret_cast<float4*>(smem);
    float* h3 = smem + 4 * PXG_SIG_NSEG;               // [2][MT][16][HS]
    float* inb = h3 + 2 * MT * 16 * HS;                // [2][MT*16][IS]
    float* bl = inb + 2 * MT * 16 * IS;                // [H][4] bias, (i,f,g,o) per unit
    int* ridx = reinterpret_cast<int*>(bl + 4 * H);    // [16*MT]
    int* s_task = ridx + 16 * MT;

    const int tid = threadIdx.x, lane = tid & 63, slice = tid >> 6;
    const int rd_l = lane & 15, ul = lane >> 4;

    load_sigtab(tab, sigtab, tid);
    for (int i = tid; i < 4 * H; i += LSTM_THREADS) bl[i] = b3[(i & 3) * H + (i >> 2)];
    float wI[NT][KBI], wR[NT][KBR];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int unit0 = slice * 16 + nt * 4;
        load_wfrag<H, KBI>(wI[nt], W3, 0, unit0, lane);
        load_wfrag<H, KBR>(wR[nt], U3, 0, unit0, lane);
    }
    constexpr int NPF = (MT * 16 * (2 * HI / 4) + LSTM_THREADS - 1) / LSTM_THREADS;

    for (;;) {
        __syncthreads();
        if (tid == 0) *s_task = atomicAdd(&queue[0], 1);
        __syncthreads();
        const int q = *s_task;
        if (q >= n_tasks) break;
        const int blk = q / n_groups, grp = q % n_groups;
        const int tile0 = grp * MT;
        const int ntile = min(MT, n_tiles - tile0);
        const int t0 = blk * QB;
        const int t1 = min(t0 + QB, T);
        for (int i = tid; i < 16 * MT; i += LSTM_THREADS) {
            const int row = tile0 * 16 + i;
            ridx[i] = (i < 16 * ntile && row < lim) ? (idx ? idx[row] : row) : -1;
        }
        float c3[MT][NT];
        if (blk == 0) {
            for (int i = tid; i < 2 * MT * 16 * HS; i += LSTM_THREADS) h3[i] = 0.0f;
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) c3[m][nt] = 0.0f;
            __syncthreads();
        } else {
            dq_wait(done, grp, blk, errflag, tid);
            const int rb = t0 & 1;
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const float* st_in = state + ((size_t)((blk - 1) & 1) * n_tiles + min(tile0 + m, n_tiles - 1)) * DQB_STATE;
                for (int i = tid; i < 16 * HS; i += LSTM_THREADS) h3[(rb * MT + m) * 16 * HS + i] = st_in[i];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) c3[m][nt] = st_in[16 * HS + nt * LSTM_THREADS + tid];
            }
            __syncthreads();
        }
        // input rows of step t0 into the buffer step t0 reads; prefetch constants of the task
        const int n_f4 = 16 * ntile * (2 * HI / 4);    // float4 per step of input rows
        const float* pf_src[NPF];
        int pf_dst[NPF];
#pragma unroll
        for (int p = 0; p < NPF; p++) {
            const int i = tid + p * LSTM_THREADS;
            pf_src[p] = nullptr;
            pf_dst[p] = -1;
            if (i < n_f4) {
                const int row = i / (2 * HI / 4), c4 = i % (2 * HI / 4);
                const int rd = ridx[row];
                pf_dst[p] = row * IS + c4 * 4;
                if (rd >= 0) pf_src[p] = bidir + ((size_t)rd * T) * (2 * HI) + c4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rd >= 0) v = *reinterpret_cast<const float4*>(pf_src[p] + (size_t)t0 * (2 * HI));
                *reinterpret_cast<float4*>(inb + (t0 & 1) * MT * 16 * IS + pf_dst[p]) = v;
            }
        }
        __syncthreads();

        for (int t = t0; t < t1; t++) {
            const int rdb = t & 1, wrb = (t + 1) & 1;
            float4 pf[NPF];
#pragma unroll
            for (int p = 0; p < NPF; p++) {
                pf[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pf_src[p] != nullptr && t + 1 < t1)
                    pf[p] = *reinterpret_cast<const float4*>(pf_src[p] + (size_t)(t + 1) * (2 * HI));
            }
#pragma unroll
            for (int m = 0; m < MT; m++) {
                if (m < ntile) {
                    f32x4 acc[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        const float4 bv = *reinterpret_cast<const float4*>(bl + (slice * 16 + nt * 4 + ul) * 4);
                        acc[nt][0] = bv.x; acc[nt][1] = bv.y; acc[nt][2] = bv.z; acc[nt][3] = bv.w;
                    }
                    const float* p = inb + (rdb * MT * 16 + m * 16) * IS + rd_l * IS + (lane >> 4) * (HI / 4);
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        float a[HI / 4];
#pragma unroll
                        for (int v4 = 0; v4 < HI / 16; v4++) {
                            const float4 v = *reinterpret_cast<const float4*>(p + half * HI + 4 * v4);
                            a[4 * v4] = v.x; a[4 * v4 + 1] = v.y; a[4 * v4 + 2] = v.z; a[4 * v4 + 3] = v.w;
                        }
#pragma unroll
                        for (int kb = 0; kb < HI / 4; kb++)
#pragma unroll
                            for (int nt = 0; nt < NT; nt++)
                                acc[nt] = mfma4(wI[nt][half * (HI / 4) + kb], a[kb], acc[nt]);
                    }
                    {
                        float a[KBR];
                        load_afrag<H>(a, h3 + (rdb * MT + m) * 16 * HS, lane);
#pragma unroll
                        for (int kb = 0; kb < KBR; kb++)
#pragma unroll
                            for (int nt = 0; nt < NT; nt++) acc[nt] = mfma4(wR[nt][kb], a[kb], acc[nt]);
                    }
                    float* o3 = h3 + ((wrb * MT + m) * 16 + rd_l) * HS;
                    float hn[NT];
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        f32x4 a2[2] = { acc[2 * half], acc[2 * half + 1] };
                        float cc[2] = { c3[m][2 * half], c3[m][2 * half + 1] };
                        float hh[2];
                        cells_update<2>(tab, a2, cc, hh);
                        c3[m][2 * half] = cc[0]; c3[m][2 * half + 1] = cc[1];
                        hn[2 * half] = hh[0]; hn[2 * half + 1] = hh[1];
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) o3[hpos<H>(slice * 16 + nt * 4 + ul)] = hn[nt];
                }
            }
#pragma unroll
            for (int p2 = 0; p2 < NPF; p2++)
                if (pf_dst[p2] >= 0)
                    *reinterpret_cast<float4*>(inb + wrb * MT * 16 * IS + pf_dst[p2]) = pf[p2];
            __syncthreads();
        }

        if (t1 == T) {
            // ---- Dense(n_classes) + softmax ------------------------------------
            const int fin = T & 1;
            for (int row = tid; row < 16 * ntile; row += LSTM_THREADS) {
                const int rd = ridx[row];
                if (rd < 0) continue;
                const float* hr = h3 + ((fin * MT + (row >> 4)) * 16 + (row & 15)) * HS;
                float z[PXG_MAX_CLASSES], e[PXG_MAX_CLASSES];
#pragma unroll
                for (int j = 0; j < PXG_MAX_CLASSES; j++) {
                    z[j] = -__builtin_inff();
                    if (j < n_classes) {
                        float acc = bd[j];
                        for (int k = 0; k < H; k++) acc = __builtin_fmaf(hr[hpos<H>(k)], Wd[k * n_classes + j], acc);
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
            const int rb = t1 & 1;
#pragma unroll
            for (int m = 0; m < MT; m++) {
                if (m < ntile) {
                    float* st_out = state + ((size_t)(blk & 1) * n_tiles + tile0 + m) * DQB_STATE;
                    for (int i = tid; i < 16 * HS; i += LSTM_THREADS) st_out[i] = h3[(rb * MT + m) * 16 * HS + i];
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) st_out[16 * HS + nt * LSTM_THREADS + tid] = c3[m][nt];
                }
            }
            dq_publish(done, grp, blk, tid);
        }
    }
}

// ===========================================================================
// launchers
// ===========================================================================
struct LstmGrid { int blocks, mtw; };

static LstmGrid pick_grid(pxg_ctx* ctx, int64_t n_rows, int maxt = LSTM_MAXT)
{
    // 2 workgroups per CU are resident.  More tiles than that: launch a whole number of
    // "rounds" of resident groups and spread the tiles evenly over them (my_tiles gives the
    // first groups one tile more), instead of filling groups to maxt and leaving a last,
    // nearly empty round: 6250 tiles (100 000 reads) = 2048 groups of 3-4 tiles, not 1563 of 4.
    const int64_t tiles = (n_rows + 15) / 16;
    const int64_t slots = 2 * (int64_t)ctx->n_cu;
    if (tiles <= slots) return { (int)tiles, 1 };
    const int64_t rounds = (tiles + slots * maxt - 1) / (slots * maxt);
    const int64_t blocks = rounds * slots;
    const int64_t mtw = (tiles + blocks - 1) / blocks;
    return { (int)blocks, (int)mtw };
}

int pxg_lstm_upload(pxg_ctx* ctx)
{
    const pxg_config& c = ctx->cfg;
    const bool ok = c.scaler_lstm1.input_dim == 1 && c.scaler_lstm1.units == 48 &&
                    c.scaler_lstm2.input_dim == 48 && c.scaler_lstm2.units == 48 &&
                    c.scaler_dense.in_dim == 48 && c.scaler_dense.out_dim == 2 &&
                    c.demux_fwd.input_dim == 1 && c.demux_fwd.units == 48 &&
                    c.demux_bwd.input_dim == 1 && c.demux_bwd.units == 48 &&
                    c.demux_top.input_dim == 96 && c.demux_top.units == 64 &&
                    c.demux_dense.in_dim == 64 && c.demux_dense.out_dim <= PXG_MAX_CLASSES &&
                    (c.scaler_length / c.stride) % 4 == 0;
    if (!ok) {
        pxg_set_err(ctx, "LSTM kernels are specialised for the MIN106-RNA001 model shapes "
                   "(scaler 1-48-48-2, demux 1-2x48-64-5)");
        return PXG_E_UNSUPPORTED;
    }
    return PXG_OK;
}

static const size_t kTabBytes = sizeof(float) * 4 * PXG_SIG_NSEG;

#define LSTM_DISPATCH(MTWVAR, CALL)  \
    switch (MTWVAR) {                \
    case 1: { CALL(1); } break;      \
    case 2: { CALL(2); } break;      \
    case 3: { CALL(3); } break;      \
    default: { CALL(4); } break;     \
    }

// error flag of the time-sliced kernels: allocated and cleared once, read back by
// pxg_batch_sync / pxg_batch_download
static int pxg_timeslice_prepare(pxg_ctx* ctx)
{
    if (!ctx->lstm_err.p) {
        int rc = pxg_reserve(ctx, ctx->lstm_err, 1);
        if (rc) return rc;
        PXG_HIP(ctx, hipMemsetAsync(ctx->lstm_err.p, 0, sizeof(int), ctx->stream));
    }
    return PXG_OK;
}

int pxg_launch_scaler_lstm(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx,
                           const int32_t* count, const float* head, float* pred, const int64_t* off)
{
    if (n_rows <= 0) return PXG_OK;
    if (ctx->cfg.lstm_arith == PXG_LSTM_Q8) return pxg_launch_scaler_lstm_q8(ctx, n_rows, idx, count, head, pred, off);
    const int T = ctx->cfg.scaler_length / ctx->cfg.stride;
    const LstmGrid g = pick_grid(ctx, n_rows);
    const PxgLstmDev &l1 = ctx->scaler1, &l2 = ctx->scaler2;
    const int64_t tiles = (n_rows + 15) / 16, slots = 2 * (int64_t)ctx->n_cu;
    if (tiles > slots && !getenv("PXG_NO_TIMESLICE")) {
        // more tiles than resident workgroups: time-slice (see k_scaler_lstm_q); measured against
        // the static split: 10 000 reads 13.3 -> 10.7 ms, 50 000 reads 75 -> 51, 100 000 reads 131 -> 101
        int rc;
        if ((rc = pxg_timeslice_prepare(ctx)) || (rc = pxg_reserve(ctx, ctx->lstm_q, (size_t)(2 + tiles))) ||
            (rc = pxg_reserve(ctx, ctx->lstm_state, (size_t)2 * tiles * QSTATE)))
            return rc;
        PXG_HIP(ctx, hipMemsetAsync(ctx->lstm_q.p, 0, (size_t)(2 + tiles) * sizeof(int), ctx->stream));
        const size_t lds = kTabBytes + sizeof(float) * (4 * 16 * HSTRIDE(48) + 16 * XS) + sizeof(int) * 32;
        PXG_HIP(ctx, hipFuncSetAttribute((const void*)k_scaler_lstm_q,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_scaler_lstm_q, dim3((unsigned)slots), dim3(LSTM_THREADS), lds, ctx->stream,
                           (int)n_rows, idx, count, T, head, ctx->d_sigtab, l1.kernel, l1.recurrent,
                           l1.bias, l2.kernel, l2.recurrent, l2.bias, ctx->scaler_dense.kernel,
                           ctx->scaler_dense.bias, pred, ctx->lstm_q.p, ctx->lstm_err.p, ctx->lstm_q.p + 2,
                           ctx->lstm_state.p);
        ctx->timeslice_used = true;
        PXG_HIP(ctx, hipGetLastError());
        return PXG_OK;
    }
#define CALL(M)                                                                                  \
    const size_t lds = kTabBytes + sizeof(float) * (4 * M * 16 * HSTRIDE(48) + 16 * M * XS) +    \
                       sizeof(int) * 16 * M;                                                     \
    PXG_HIP(ctx, hipFuncSetAttribute((const void*)k_scaler_lstm<M>,                              \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
    hipLaunchKernelGGL(k_scaler_lstm<M>, dim3(g.blocks), dim3(LSTM_THREADS), lds, ctx->stream,   \
                       (int)n_rows, idx, count, T, head, ctx->d_sigtab, l1.kernel, l1.recurrent, \
                       l1.bias, l2.kernel, l2.recurrent, l2.bias, ctx->scaler_dense.kernel,      \
                       ctx->scaler_dense.bias, pred);
    LSTM_DISPATCH(g.mtw, CALL)
#undef CALL
    PXG_HIP(ctx, hipGetLastError());
    return PXG_OK;
}

int pxg_launch_demux_lstm(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx,
                          const int32_t* count, const float* win, float* bidir, float* probs,
                          int timer_a, int timer_b)
{
    if (n_rows <= 0) return PXG_OK;
    if (ctx->cfg.lstm_arith == PXG_LSTM_Q8)
        return pxg_launch_demux_lstm_q8(ctx, n_rows, idx, count, win, bidir, probs, timer_a, timer_b);
    const int T = ctx->cfg.signal_trim_length;
    const LstmGrid g = pick_grid(ctx, n_rows);
    const int64_t tiles = (n_rows + 15) / 16, slots = 2 * (int64_t)ctx->n_cu;
    if (tiles > slots && !getenv("PXG_NO_TIMESLICE") && !getenv("PXG_NO_DEMUX_TIMESLICE")) {
        // more tiles than resident workgroups: time-sliced kernels (k_demux_bidir_q / k_demux_top_q)
        const size_t state_floats = (size_t)2 * tiles * (DQA_STATE > DQB_STATE ? DQA_STATE : DQB_STATE);
        int rc;
        if ((rc = pxg_timeslice_prepare(ctx)) || (rc = pxg_reserve(ctx, ctx->demux_q, (size_t)2 * (2 + tiles))) ||
            (rc = pxg_reserve(ctx, ctx->demux_state, state_floats)))
            return rc;
        PXG_HIP(ctx, hipMemsetAsync(ctx->demux_q.p, 0, (size_t)2 * (2 + tiles) * sizeof(int), ctx->stream));
        int* qa = ctx->demux_q.p;
        int* qb = ctx->demux_q.p + 2 + tiles;
        if (const char* forced = getenv("PXG_DEMUX_BLOCKS")) {      // tuning knob: step blocks per tile
            static int nb_a, nb_b;                                   // "a,b" or one number for both kernels
            nb_a = nb_b = atoi(forced);
            for (const char* c = forced; *c; c++)
                if (*c == ',') nb_b = atoi(c + 1);
            PXG_HIP(ctx, hipMemcpyAsync(qa + 1, &nb_a, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            PXG_HIP(ctx, hipMemcpyAsync(qb + 1, &nb_b, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        }
        const PxgLstmDev &f = ctx->demux_fwd, &b = ctx->demux_bwd, &t3 = ctx->demux_top;
#define CALL_A(MT)                                                                                         \
    {                                                                                                      \
        const size_t lds = kTabBytes + sizeof(float) * (4 * MT * 16 * HSTRIDE(48) + 2 * 16 * MT * XS) +    \
                           sizeof(int) * (16 * MT + 16);                                                   \
        PXG_HIP(ctx, hipFuncSetAttribute((const void*)k_demux_bidir_q<MT>,                                 \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
        hipLaunchKernelGGL(k_demux_bidir_q<MT>, dim3((unsigned)slots), dim3(LSTM_THREADS), lds,            \
                           ctx->stream, (int)n_rows, idx, count, T, win, ctx->d_sigtab, f.kernel,          \
                           f.recurrent, f.bias, b.kernel, b.recurrent, b.bias, bidir, qa, ctx->lstm_err.p, \
                           qa + 2, ctx->demux_state.p);                                                    \
    }
#define CALL_B(MT)                                                                                         \
    {                                                                                                      \
        const size_t lds = kTabBytes + sizeof(float) * (2 * MT * 16 * HSTRIDE(64) +                        \
                                                        2 * MT * 16 * HSTRIDE(96) + 256) +                 \
                           sizeof(int) * (16 * MT + 16);                                                   \
        PXG_HIP(ctx, hipFuncSetAttribute((const void*)k_demux_top_q<MT>,                                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));           \
        hipLaunchKernelGGL(k_demux_top_q<MT>, dim3((unsigned)slots), dim3(LSTM_THREADS), lds, ctx->stream, \
                           (int)n_rows, idx, count, T, bidir, ctx->d_sigtab, t3.kernel, t3.recurrent,      \
                           t3.bias, ctx->demux_dense.kernel, ctx->demux_dense.bias,                        \
                           ctx->demux_dense.out_dim, probs, qb, ctx->lstm_err.p, qb + 2,                   \
                           ctx->demux_state.p);                                                            \
    }
        pxg_timer_begin(ctx, timer_a);
        CALL_A(1)
        pxg_timer_end(ctx, timer_a);
        pxg_timer_begin(ctx, timer_b);
        CALL_B(1)
        pxg_timer_end(ctx, timer_b);
#undef CALL_A
#undef CALL_B
        ctx->timeslice_used = true;
        PXG_HIP(ctx, hipGetLastError());
        return PXG_OK;
    }
    {
        const PxgLstmDev &f = ctx->demux_fwd, &b = ctx->demux_bwd;
        pxg_timer_begin(ctx, timer_a);
#define CALL(M)                                                                                  \
    const size_t lds = kTabBytes + sizeof(float) * (4 * M * 16 * HSTRIDE(48) + 2 * 16 * M * XS) +         \
                       sizeof(int) * 16 * M;                                                     \
    PXG_HIP(ctx, hipFuncSetAttribute((const void*)k_demux_bidir<M>,                              \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
    hipLaunchKernelGGL(k_demux_bidir<M>, dim3(g.blocks), dim3(LSTM_THREADS), lds, ctx->stream,   \
                       (int)n_rows, idx, count, T, win, ctx->d_sigtab, f.kernel, f.recurrent,    \
                       f.bias, b.kernel, b.recurrent, b.bias, bidir);
        LSTM_DISPATCH(g.mtw, CALL)
#undef CALL
        pxg_timer_end(ctx, timer_a);
    }
    {
        const PxgLstmDev& t3 = ctx->demux_top;
        const LstmGrid g = pick_grid(ctx, n_rows, 2);   // 160 weight VGPRs: 2 tiles max
        pxg_timer_begin(ctx, timer_b);
#define CALL(M)                                                                                  \
    const size_t lds = kTabBytes + sizeof(float) * (2 * M * 16 * HSTRIDE(64) + 2 * M * 16 * HSTRIDE(96) + 256) +   \
                       sizeof(int) * 16 * M;                                                     \
    PXG_HIP(ctx, hipFuncSetAttribute((const void*)k_demux_top<M>,                                \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
    hipLaunchKernelGGL(k_demux_top<M>, dim3(g.blocks), dim3(LSTM_THREADS), lds, ctx->stream,     \
                       (int)n_rows, idx, count, T, bidir, ctx->d_sigtab, t3.kernel,              \
                       t3.recurrent, t3.bias, ctx->demux_dense.kernel, ctx->demux_dense.bias,    \
                       ctx->demux_dense.out_dim, probs);
        if (g.mtw == 1) { CALL(1); } else { CALL(2); }
#undef CALL
        pxg_timer_end(ctx, timer_b);
    }
    PXG_HIP(ctx, hipGetLastError());
    return PXG_OK;
}
