// k_lstm_shared.h -- pieces shared by the two arithmetics of the recurrent networks
// (k_lstm.hip: float32 fma chains on v_mfma_f32_16x16x4_f32; k_lstm_q8.hip: exact fixed point on
// v_mfma_i32_16x16x64_i8): the activation / cell update on pre-activations in table units, and the
// task-queue helpers of the time-sliced kernels.
#pragma once
#include "pxg_common.h"

#define LSTM_THREADS 256
#define LSTM_MAXT 4     // M-tiles per workgroup
#define XCH 64          // scaler: steps of x staged per LDS refill
#define XS (XCH + 4)    // padded row stride (floats)


// acc = pre-activations of one (read, unit), ALREADY in table units: the gate
// columns of every LSTM matrix/bias are scaled at upload (pxg_api.hip
// upload_lstm) by 16 (i, f, o: sigmoid table step 1/16) and 32 (g: tanh =
// 2*sigmoid(2x)-1).  A power-of-two scale commutes with every rounding of the
// fma chain, so acc == 16z resp. 32z bit for bit and the lookups lose their
// multiply.  The cell state is carried as C = 32c for the same reason:
// fl(f*C) + fl(i*G) with G = fma(64, s, -32) = 32*tanh is exactly 32*c'.
// Returns h = fl(o * tanh(c')), identical to the unscaled formulation.
__device__ __forceinline__ float cell_update(const float4* tab, f32x4 acc, float& C)
{
    const float ig = pxg_sig_lookup_u(tab, acc[0]);
    const float fg = pxg_sig_lookup_u(tab, acc[1]);
    const float G = __builtin_fmaf(64.0f, pxg_sig_lookup_u(tab, acc[2]), -32.0f);
    const float og = pxg_sig_lookup_u(tab, acc[3]);
    const float fc = fg * C;
    const float in = ig * G;
    const float cn = fc + in;
    C = cn;
    const float th = __builtin_fmaf(2.0f, pxg_sig_lookup_u(tab, cn), -1.0f);
    return og * th;
}

// The same update for the NT cells a lane owns in one layer, written in stages so
// that the NT*4 table rows are requested back to back (one exposed LDS latency per
// layer and tile instead of one per activation).
template <int NT>
__device__ __forceinline__ void cells_update(const float4* tab, const f32x4 (&acc)[NT], float (&C)[NT],
                                             float (&h)[NT])
{
    float s[NT][4];
    float4 c[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float u = __builtin_amdgcn_fmed3f(acc[nt][r], -512.0f, 511.99997f);
            s[nt][r] = pxg_sig_position(u);
            c[nt][r] = tab[pxg_sig_segment(u) + PXG_SIG_HALF];
        }
    float s2[NT], og[NT];
    float4 c2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        float g[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float p = __builtin_fmaf(c[nt][r].w, s[nt][r], c[nt][r].z);
            p = __builtin_fmaf(p, s[nt][r], c[nt][r].y);
            g[r] = __builtin_fmaf(p, s[nt][r], c[nt][r].x);
        }
        const float G = __builtin_fmaf(64.0f, g[2], -32.0f);
        const float fc = g[1] * C[nt];
        const float in = g[0] * G;
        const float cn = fc + in;
        C[nt] = cn;
        og[nt] = g[3];
        const float u = __builtin_amdgcn_fmed3f(cn, -512.0f, 511.99997f);
        s2[nt] = pxg_sig_position(u);
        c2[nt] = tab[pxg_sig_segment(u) + PXG_SIG_HALF];
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        float p = __builtin_fmaf(c2[nt].w, s2[nt], c2[nt].z);
        p = __builtin_fmaf(p, s2[nt], c2[nt].y);
        const float st = __builtin_fmaf(p, s2[nt], c2[nt].x);
        const float th = __builtin_fmaf(2.0f, st, -1.0f);
        h[nt] = og[nt] * th;
    }
}

__device__ __forceinline__ void load_sigtab(float4* dst, const float* __restrict__ src, int tid)
{
    for (int i = tid; i < PXG_SIG_NSEG; i += 256)
        dst[i] = reinterpret_cast<const float4*>(src)[i];
}


#define SQ_MAXBLK 32            // most step blocks per tile
#define SQ_HANDOVER 3           // cost of one hand-over, in steps (cost model only)

// Steps per task, chosen on the device from the actual tile count: the launch takes
// ceil(tiles x blocks / slots) rounds of tasks, so the cut is the block count that minimises
// rounds x block length.  (Round 1 used a fixed 256 steps: 625 tiles x 8 blocks on 512 slots
// = 9.77 rounds = 9 x 256 + 209 = 2 513 step times for 2 442 of work; 9 blocks of 224 steps
// fill 10.99 rounds: 2 449.)  Lengths are multiples of 4 so that the float4 loads of the x
// tile stay aligned.
__device__ __forceinline__ int scaler_block_steps(int n_tiles, int slots, int T1)
{
    int best = T1, best_cost = 0x7fffffff;
    for (int nb = 1; nb <= SQ_MAXBLK; nb++) {
        const int qb = (((T1 + nb - 1) / nb) + 3) & ~3;
        const int blocks = (T1 + qb - 1) / qb;
        // (the blocks of ONE tile run one after the other: with fewer tiles than slots the chain, not the rounds, is
        //  the launch's length -- a small batch takes its tiles whole)
        const int rounds = max((blocks * n_tiles + slots - 1) / slots, blocks);
        const int cost = rounds * (qb + SQ_HANDOVER);
        if (cost < best_cost) { best_cost = cost; best = qb; }
    }
    return best;
}

#define DQ_HANDOVER 24          // cost of one hand-over, in steps (cost model only)
#define DQ_MAXBLK 12

// Step blocks per tile.  Measured (PXG_DEMUX_BLOCKS sweep, 5 000 - 100 000 reads, profiles/r02):
// every extra block per tile costs 25 - 40 us of hand-over per slot (state through HBM, L2
// write-back / invalidate at agent scope with the XCD's L2 full of the streamed rows), and a last
// round of tasks that fills at most half of the slots runs about twice as fast (one workgroup per
// CU issues alone).  So: one block per tile unless cutting removes a nearly empty round --
// 10 000 reads = 619 tiles on 512 slots: 2 blocks (2.4 rounds of 150 steps: K5a 1.56 -> 1.33 ms,
// K5b 2.28 -> 2.12; the first model, whole rounds and 3 steps per hand-over, chose 4);
// 20 000 - 100 000 reads: 1 block.
__device__ __forceinline__ int demux_blocks(int n_tiles, int slots, int T)
{
    int best = 1;
    float best_cost = 3.0e38f;
    for (int nb = 1; nb <= DQ_MAXBLK; nb++) {
        const int tasks = nb * n_tiles;
        const int full = tasks / slots, rem = tasks - full * slots;
        const float rounds = fmaxf((float)full + (rem == 0 ? 0.0f : (2 * rem <= slots ? 0.5f : 1.0f)),
                                   nb > 1 ? (float)nb : 0.0f);      // a tile's blocks are a chain
        const float cost = rounds * (float)((T + nb - 1) / nb) + (float)((nb - 1) * DQ_HANDOVER);
        if (cost < 0.96f * best_cost) { best_cost = cost; best = nb; }      // more blocks must pay clearly
    }
    return best;
}

// wait until block `blk` of `tile` may start (its predecessor published), bounded spin
__device__ __forceinline__ void dq_wait(const int* done, int tile, int blk, int* errflag, int tid)
{
    if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(&done[tile], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < blk) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1 << 24)) { atomicExch(errflag, 1); break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__device__ __forceinline__ void dq_publish(int* done, int tile, int blk, int tid)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&done[tile], blk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

