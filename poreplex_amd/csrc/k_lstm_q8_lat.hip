// k_lstm_q8_lat.hip -- the LATENCY form of K2 (scaler LSTM, PXG_LSTM_Q8): small batches.
//
// Why.  k_scaler_lstm_q8 gives a 16-read tile to one 4-wave workgroup; its SIMDs issue 63 MFMAs + ~460 VALU per wave
// and step whatever the batch size, so a batch below 16 x #CU reads runs its 2 000 serial steps at ~3 800 cycles
// each on a fraction of the chip (a merged group of 128-read process_batch calls: ~60 of 256 CUs, two thirds of its
// time).  More waves per tile on the SAME CU change little -- the four SIMDs of the CU are the bound.  The form below
// spreads a small batch over FOUR TIMES as many CUs: a tile is 4 reads, and the twelve idle read columns of the
// 16x16x64 MFMA carry the digit planes instead:
//
//   B column n = 4 s + rd (s = slot 0-3, rd = read 0-3); one k group of a hidden vector in LDS is a strip of six
//   64-byte positions [0 | 0 | h2 | h1 | h0 | 0] (h2 = top digit plane, 4 reads x 16 k bytes each); the B fragment of
//   weight digit w2 starts at position 2, of w1 at position 1, of w0 at position 0:
//       w2 x [h2 h1 h0 0 ]   w1 x [0  h2 h1 h0]   w0 x [0  0  h2 h1]
//   so THREE MFMAs into ONE accumulator leave significance level l (A_l of oracle/pxo_core.c lstm_step_q; w0 h0 is
//   dropped by construction) in slot l -- the same integers the 8-product form sums, in 3/8 of the instructions and a
//   quarter of the accumulators.  Layer 2's second k block holds 32 of its 96 inputs: its two halves carry (w2, w1)
//   with the fragment read at positions (2, 1), then (w0, 0): 5 MFMAs per layer-2 gate tile, 3 per layer-1 tile.
//   Levels -> float inside the 16-lane row (DPP): W = (acc << 8)[lane - 4] + acc is V = A0 256 + A1 in slot 1 and
//   U = A2 256 + A3 in slot 3; the W of a second gate tile, rotated by four lanes, fills slots 2 and 0 of the same
//   register; F = (float)W; t = fma(F[lane - 8 mod 16], 65536, F) is the pre-activation integer of tile 0 in slot 3
//   and of tile 1 in slot 0; a third tile's t moves from slot 3 to slot 2.  Lane (a, slot, rd) then holds unit a of
//   the gate tile of its slot for read rd: the gate math (30 of K2's 77 VALU per tile) runs once per THREE tiles.
//   A workgroup is 8 waves: waves 0-3 own layer 1 (12 units each, 9 MFMAs), waves 4-7 layer 2 (15 MFMAs) -- two
//   waves per SIMD, each hiding the other's LDS / DPP / MFMA latencies.
//
// Same integers, same float32 operations per cell as k_scaler_lstm_q8 and the oracle: bit-exact by construction
// (tests/test_gpu_parity.py scaler hooks run this kernel for every n <= 8 x #CU).  No time-slicing and no zero-pad
// prefix table: every tile is resident and takes all T + 1 iterations.
#include <type_traits>
#include "k_lstm_q8.h"

#define QL_STRIP 384                    // one k group of one hidden vector: [6 positions][4 reads][16 bytes]
#define QL_HV (6 * QL_STRIP)            // h1 groups 0-2 (strips 0-2), h2 groups 0-2 (strips 3-5)
#define QL_FRAGS 8                      // per gate tile: L1 w2 w1 w0 | L2 block 0 w2 w1 w0 | L2 block 1' [w2 w2 w1 w1], [w0 w0 0 0]
#define QL_XS (XCH + 4)
#define QL_THREADS 512

#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_ROR(n) (0x120 + (n))

// (Tried: the w1 / w0 fragments as DPP-shifted copies of the w2 fragment -- the lane's slot then holds what slot - N
//  of it holds, zero-filled -- instead of further ds_read_b128 of the same bytes: K2 1.58 -> 1.75 ms at 1 024 reads, the
//  last fragment of the half block alone 1.58 -> 1.65 (profiles/r06/ab_lat_forms.txt).  The step is bound by its
//  dependent chain, not by LDS bandwidth: independent reads arrive back to back, the moves wait for the first.)

// level l in slot l of every 16-lane row -> V = A0 256 + A1 in slot 1, U = A2 256 + A3 in slot 3
__device__ __forceinline__ int ql_words(int acc)
{
    const int X = (int)((unsigned)acc << 8);
    return __builtin_amdgcn_update_dpp(0, X, DPP_ROW_SHR(4), 0xF, 0xF, true) + acc;
}

// three accumulators of one gate row (4 units x 16 columns each) -> the pre-activation integers t = fma((float)V, 65536,
// (float)U) of tile 0 in slot 3, tile 1 in slot 0, tile 2 in slot 2 (slot 1: don't care)
__device__ __forceinline__ float ql_levels3(int a0, int a1, int a2, float k65536)
{
    int r1 = ql_words(a0);
    r1 = __builtin_amdgcn_update_dpp(r1, ql_words(a1), DPP_ROW_ROR(4), 0xF, 0x5, false);     // (V, U): slots (1, 3) -> (2, 0)
    const float f1 = (float)r1, f2 = (float)ql_words(a2);
    const float v1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f1), DPP_ROW_ROR(8), 0xF, 0xF, false));
    const float v2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f2), DPP_ROW_SHR(8), 0xF, 0xF, true));
    const float t1 = __builtin_fmaf(v1, k65536, f1);
    const float t2 = __builtin_fmaf(v2, k65536, f2);
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(t1), __float_as_int(t2), DPP_ROW_SHL(4), 0xF, 0x4, false));
}

// two accumulators of one gate row -> t of tile 0 in slot 3, of tile 1 in slot 0 (slots 1, 2: don't care)
__device__ __forceinline__ float ql_levels2(int a0, int a1, float k65536)
{
    int r1 = ql_words(a0);
    r1 = __builtin_amdgcn_update_dpp(r1, ql_words(a1), DPP_ROW_ROR(4), 0xF, 0x5, false);
    const float f1 = (float)r1;
    const float v1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f1), DPP_ROW_ROR(8), 0xF, 0xF, false));
    return __builtin_fmaf(v1, k65536, f1);
}

template <int NTW>
__device__ __forceinline__ float ql_levels(const v4i (&acc)[NTW], int r, float k65536)
{
    if constexpr (NTW == 3) return ql_levels3(acc[0][r], acc[1][r], acc[2][r], k65536);
    else return ql_levels2(acc[0][r], acc[1][r], k65536);
}

// NTW: gate tiles per wave -- 3 (8 waves: 4 per layer; slots 3, 0, 2) or 2 (12 waves: 6 per layer; slots 3, 0)
template <int NTW>
__global__ __launch_bounds__(NTW == 3 ? 512 : 768) void k_scaler_lstm_q8_lat(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const float* __restrict__ head, const float* __restrict__ sigtab, const v4i* __restrict__ frag,
    const float* __restrict__ W1 /* scalar-input kernel, table units */, const float* __restrict__ b1,
    const float* __restrict__ b2, Q8Scale s1, Q8Scale s2, const float* __restrict__ Wd, const float* __restrict__ bd,
    float* __restrict__ pred)
{
    constexpr int H = 48;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    const int row_base = (int)blockIdx.x * 4;
    if (row_base >= lim) return;

    float4* tab = reinterpret_cast<float4*>(smem);                                     // [1024] sigmoid spline
    unsigned char* hv = reinterpret_cast<unsigned char*>(smem + 4 * PXG_SIG_NSEG);     // [2 buffers][QL_HV]
    float* xb = reinterpret_cast<float*>(hv + 2 * QL_HV);                              // [2 chunks][4 reads][QL_XS]
    int* ridx = reinterpret_cast<int*>(xb + 2 * 4 * QL_XS);                            // [4]

    constexpr int WPL = 12 / NTW, THREADS = 2 * WPL * 64;            // waves per layer, threads per workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const bool l1 = wv < WPL;                  // the first WPL waves: layer 1, the others: layer 2 (wave-uniform: scalar branches)
    const int wl = l1 ? wv : wv - WPL;         // this wave's gate tiles of its layer: NTW * wl + i (tile g = units 4 g .. 4 g + 3)
    const int rd = lane & 3, slot = (lane >> 2) & 3, a = lane >> 4, kg = lane >> 4;

    for (int i = tid; i < PXG_SIG_NSEG; i += THREADS) tab[i] = reinterpret_cast<const float4*>(sigtab)[i];
    v4i wq[NTW][5];                             // weight fragments of this wave's gate tiles, in VGPRs throughout
#pragma unroll
    for (int i = 0; i < NTW; i++) {
        const int g = NTW * wl + i;             // (the fragments are laid out for 4 waves x 3 tiles: tile g = wave g / 3, nt g % 3)
        const v4i* f = frag + (size_t)(g % 3) * QL_FRAGS * LSTM_THREADS + (g / 3) * 64 + lane;
#pragma unroll
        for (int d = 0; d < 5; d++) wq[i][d] = f[(l1 ? (d > 2 ? 2 : d) : 3 + d) * LSTM_THREADS];
    }
    for (int i = tid; i < 2 * QL_HV / 4; i += THREADS) reinterpret_cast<unsigned*>(hv)[i] = 0u;
    if (tid < 4) {
        const int row = row_base + tid;
        ridx[tid] = row < lim ? (idx ? idx[row] : row) : -1;
    }
    __syncthreads();

    // the gate tile a slot carries after ql_levels3 / ql_levels2: slot 3 -> tile 0, slot 0 -> tile 1, slot 2 -> tile 2 (NTW = 3)
    const int nt_of_slot = slot == 3 ? 0 : (slot == 0 ? 1 : 2);
    const bool publishes = NTW == 3 ? slot != 1 : (slot == 3 || slot == 0);
    const int unit = 4 * (NTW * wl + (nt_of_slot < NTW ? nt_of_slot : 0)) + a;
    float stW[4], stB[4], sc[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        stW[r] = l1 ? W1[r * H + unit] : 0.0f;
        stB[r] = l1 ? b1[r * H + unit] : b2[r * H + unit];
        sc[r] = l1 ? s1.g[r] : s2.g[r];
    }
    // byte offsets inside a hidden-vector buffer
    const int offF = kg * QL_STRIP + (lane & 15) * 16;                                          // block 0, + 64 x first position
    const int off4 = (4 + (kg & 1)) * QL_STRIP + (2 - (kg >> 1)) * 64 + (lane & 15) * 16;      // block 1': (w2 | w1) halves
    const int off5 = (4 + (kg & 1)) * QL_STRIP + (lane & 15) * 16;                             // block 1': (w0 | 0)
    const int pub = ((l1 ? 0 : 3) + (unit >> 4)) * QL_STRIP + rd * 16 + (unit & 15);           // + 64 x (4 - digit)

    // x: one float per thread of waves 0-3 and 64-step chunk, the next chunk in flight under the current one
    const bool xs = tid < 256;                  // (the first four waves stage x: layer-1 waves in both forms)
    const int xrow = (tid >> 6) & 3, xcol = tid & 63;
    const int xrd = ridx[xrow];
    const float* xsrc = head + (size_t)(xrd < 0 ? 0 : xrd) * T;
    if (xs) xb[xrow * QL_XS + xcol] = (xrd >= 0 && xcol < T) ? xsrc[xcol] : 0.0f;
    float xnext = 0.0f;
    __syncthreads();

    float C = 0.0f;
    if (!l1) __builtin_amdgcn_s_setprio(2);      // the layer-2 wave has the longer chain (15 against 9 MFMAs): it issues first when both are ready (-3 % at 128 reads)
    float k65536;                              // (kept in a register: the DPP forms take no literal)
    asm volatile("v_mov_b32 %0, 0x47800000" : "=v"(k65536));
    const v4i z = {0, 0, 0, 0};
    // one iteration; ACT: this wave's layer runs a step in it (layer 1: t < T, layer 2: t >= 1 -- wave-uniform, and
    // false only in the first / last iteration, which are peeled so that the T - 1 steady-state steps carry no select)
    auto step = [&](int t, auto act_c) {
        constexpr bool ACT = decltype(act_c)::value;
        const unsigned char* hr = hv + (t & 1) * QL_HV;
        unsigned char* hw = hv + ((t + 1) & 1) * QL_HV;
        const int tc = t & (XCH - 1);
        if (xs && tc == 0) xnext = (xrd >= 0 && t + XCH + xcol < T) ? xsrc[t + XCH + xcol] : 0.0f;

        const v4i f2 = *reinterpret_cast<const v4i*>(hr + offF + 128);
        const v4i f1 = *reinterpret_cast<const v4i*>(hr + offF + 64);
        const v4i f0 = *reinterpret_cast<const v4i*>(hr + offF);
        v4i acc[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) acc[nt] = mfma8(wq[nt][0], f2, z);
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) acc[nt] = mfma8(wq[nt][1], f1, acc[nt]);
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) acc[nt] = mfma8(wq[nt][2], f0, acc[nt]);
        f32x4 u[1];
        if (l1) {
            const float x = xb[((t / XCH) & 1) * 4 * QL_XS + rd * QL_XS + tc];
#pragma unroll
            for (int r = 0; r < 4; r++)
                u[0][r] = __builtin_fmaf(ql_levels<NTW>(acc, r, k65536), sc[r],
                                         __builtin_fmaf(x, stW[r], stB[r]));
        } else {
            const v4i f4 = *reinterpret_cast<const v4i*>(hr + off4);
            const v4i f5 = *reinterpret_cast<const v4i*>(hr + off5);
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) acc[nt] = mfma8(wq[nt][3], f4, acc[nt]);
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) acc[nt] = mfma8(wq[nt][4], f5, acc[nt]);
#pragma unroll
            for (int r = 0; r < 4; r++)
                u[0][r] = __builtin_fmaf(ql_levels<NTW>(acc, r, k65536), sc[r], stB[r]);
        }
        unsigned q = 0u;                        // an idle cell keeps its state and publishes zeros
        if (ACT) {
            float Cn[1] = {C}, hn[1];
            cells_update<1>(tab, u, Cn, hn);
            C = Cn[0];
            q = q8_biased(hn[0]) ^ 0x00808080u;
        }
        if (publishes) {
            hw[pub + 128] = (unsigned char)(q >> 16);
            hw[pub + 192] = (unsigned char)(q >> 8);
            hw[pub + 256] = (unsigned char)q;
        }
        if (xs && tc == XCH - 1) xb[(((t / XCH) + 1) & 1) * 4 * QL_XS + xrow * QL_XS + xcol] = xnext;
        __syncthreads();
    };
    // layer 1 runs step t in iteration t (t < T), layer 2 step t - 1 (t >= 1)
    if (l1) step(0, std::true_type()); else step(0, std::false_type());
    for (int t = 1; t < T; t++) step(t, std::true_type());
    if (T >= 1) { if (l1) step(T, std::false_type()); else step(T, std::true_type()); }

    // ---- Dense(2): fma chain over k = 0..47 from the bias, on h = q * 2^-22 ----------
    const unsigned char* hf = hv + ((T + 1) & 1) * QL_HV;
    if (tid < 8) {
        const int row = tid >> 1, o = tid & 1;
        const int rdg = ridx[row];
        if (rdg >= 0) {
            float acc = bd[o];
            for (int k = 0; k < H; k++) {
                const unsigned char* p = hf + (3 + (k >> 4)) * QL_STRIP + row * 16 + (k & 15);
                const int q = (int)(signed char)p[256] + 256 * (int)(signed char)p[192] + 65536 * (int)(signed char)p[128];
                acc = __builtin_fmaf((float)q * (1.0f / 4194304.0f), Wd[k * 2 + o], acc);
            }
            pred[(size_t)rdg * 2 + o] = acc;
        }
    }
}

// ===========================================================================
// K5a, latency form: the bidirectional layer of the demux net on 4-read tiles.  Waves 0-3 run the forward cell at
// time t, waves 4-7 the backward cell at time T-1-t -- each the layer-1 wave of the kernel above (12 units, three gate
// tiles, 9 MFMAs).  A buffer holds seven strips: hf groups 0-2, hb groups 0-2 and one of padding (the fourth k group
// of a 48-unit block carries zero weights, its bytes only have to exist).  The digit positions of the strips written
// in a step (192 of 384 bytes each) go to HBM as they are: bidir[tile][time][direction][3 strips][192] is what
// k_demux_top_q8_lat copies back into its strips.  Fragments: [direction][gate tile][w2, w1, w0][thread].
// ===========================================================================
#define QL_SEQ (3 * 192)                // bytes per tile, time step and direction in HBM

__global__ __launch_bounds__(QL_THREADS) void k_demux_bidir_q8_lat(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const float* __restrict__ win, const float* __restrict__ sigtab, const v4i* __restrict__ frag,
    const float* __restrict__ Wf, const float* __restrict__ bf, const float* __restrict__ Wb,
    const float* __restrict__ bb, Q8Scale sf, Q8Scale sb, unsigned char* __restrict__ bidir)
{
    constexpr int H = 48;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    const int row_base = (int)blockIdx.x * 4;
    if (row_base >= lim) return;

    float4* tab = reinterpret_cast<float4*>(smem);
    unsigned char* hv = reinterpret_cast<unsigned char*>(smem + 4 * PXG_SIG_NSEG);     // [2 buffers][7 strips]
    float* xw = reinterpret_cast<float*>(hv + 2 * 7 * QL_STRIP);                       // [4 reads][T]
    int* ridx = reinterpret_cast<int*>(xw + 4 * T);

    const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3;
    const bool bwd = tid >= 256;
    const int ftid = tid & 255;
    const int rd = lane & 3, slot = (lane >> 2) & 3, a = lane >> 4, kg = lane >> 4;

    for (int i = tid; i < PXG_SIG_NSEG; i += QL_THREADS) tab[i] = reinterpret_cast<const float4*>(sigtab)[i];
    v4i wq[3][3];
#pragma unroll
    for (int nt = 0; nt < 3; nt++)
#pragma unroll
        for (int d = 0; d < 3; d++) wq[nt][d] = frag[(((bwd ? 3 : 0) + nt) * 3 + d) * LSTM_THREADS + ftid];
    for (int i = tid; i < 2 * 7 * QL_STRIP / 4; i += QL_THREADS) reinterpret_cast<unsigned*>(hv)[i] = 0u;
    if (tid < 4) {
        const int row = row_base + tid;
        ridx[tid] = row < lim ? (idx ? idx[row] : row) : -1;
    }
    __syncthreads();
    for (int i = tid; i < 4 * T; i += QL_THREADS) {
        const int row = i / T, c = i - row * T;
        const int rdg = ridx[row];
        xw[i] = rdg >= 0 ? win[(size_t)rdg * T + c] : 0.0f;
    }

    const int nt_of_slot = slot == 3 ? 0 : (slot == 0 ? 1 : 2);
    const bool publishes = slot != 1;
    const int unit = 12 * w + 4 * nt_of_slot + a;
    float stW[4], stB[4], sc[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        stW[r] = (bwd ? Wb : Wf)[r * H + unit];
        stB[r] = (bwd ? bb : bf)[r * H + unit];
        sc[r] = bwd ? sb.g[r] : sf.g[r];
    }
    const int offF = ((bwd ? 3 : 0) + kg) * QL_STRIP + (lane & 15) * 16;
    const int pub = ((bwd ? 3 : 0) + (unit >> 4)) * QL_STRIP + rd * 16 + (unit & 15);
    // streaming: 72 threads move one 16-byte piece each per step
    const int sdir = tid / 36, spiece = tid % 36;
    const int s_lds = (sdir * 3 + spiece / 12) * QL_STRIP + 128 + (spiece % 12) * 16;
    unsigned char* tile_out = bidir + (size_t)blockIdx.x * T * (2 * QL_SEQ);
    __syncthreads();

    float C = 0.0f;
    float k65536;
    asm volatile("v_mov_b32 %0, 0x47800000" : "=v"(k65536));
    const v4i z = {0, 0, 0, 0};
    for (int t = 0; t <= T; t++) {
        const unsigned char* hr = hv + (t & 1) * 7 * QL_STRIP;
        unsigned char* hw = hv + ((t + 1) & 1) * 7 * QL_STRIP;
        if (t > 0 && tid < 72) {        // the planes written in the previous step: forward time t-1, backward time T-t
            const f32x4 v = *reinterpret_cast<const f32x4*>(hr + s_lds);
            const int time = sdir ? T - t : t - 1;
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(tile_out + ((size_t)time * 2 + sdir) * QL_SEQ + spiece * 16));
        }
        if (t == T) break;
        const v4i f2 = *reinterpret_cast<const v4i*>(hr + offF + 128);
        const v4i f1 = *reinterpret_cast<const v4i*>(hr + offF + 64);
        const v4i f0 = *reinterpret_cast<const v4i*>(hr + offF);
        v4i acc[3];
#pragma unroll
        for (int nt = 0; nt < 3; nt++) acc[nt] = mfma8(wq[nt][0], f2, z);
#pragma unroll
        for (int nt = 0; nt < 3; nt++) acc[nt] = mfma8(wq[nt][1], f1, acc[nt]);
#pragma unroll
        for (int nt = 0; nt < 3; nt++) acc[nt] = mfma8(wq[nt][2], f0, acc[nt]);
        const float x = xw[rd * T + (bwd ? T - 1 - t : t)];
        f32x4 u[1];
#pragma unroll
        for (int r = 0; r < 4; r++)
            u[0][r] = __builtin_fmaf(ql_levels3(acc[0][r], acc[1][r], acc[2][r], k65536), sc[r],
                                     __builtin_fmaf(x, stW[r], stB[r]));
        float Cn[1] = {C}, hn[1];
        cells_update<1>(tab, u, Cn, hn);
        C = Cn[0];
        const unsigned q = q8_biased(hn[0]) ^ 0x00808080u;
        if (publishes) {
            hw[pub + 128] = (unsigned char)(q >> 16);
            hw[pub + 192] = (unsigned char)(q >> 8);
            hw[pub + 256] = (unsigned char)q;
        }
        __syncthreads();
    }
}

// ===========================================================================
// K5b, latency form: the top cell (64 units, inputs hf | hb of K5a) + Dense + softmax on 4-read tiles.  Eight waves,
// two gate tiles each (wave W: units 16 (W >> 1) + 8 (W & 1) .. + 8).  A buffer holds ten strips: hf groups 0-2, hb
// groups 0-2 (the step's input, copied from HBM one step ahead) and the cell's own h groups 0-3.  k blocks: [hf0 hf1
// hf2 hb0], [hb1 hb2 h0 h1] (three MFMAs each) and the half block [h2 h3] as (w2 | w1), (w0 | 0): 8 MFMAs per tile.
// Fragments: [gate tile 0-3][8][thread of the 4-wave layout].
// ===========================================================================
__global__ __launch_bounds__(QL_THREADS) void k_demux_top_q8_lat(
    int n_rows, const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int T,
    const unsigned char* __restrict__ bidir, const float* __restrict__ sigtab, const v4i* __restrict__ frag,
    const float* __restrict__ b3, Q8Scale s3, const float* __restrict__ Wd, const float* __restrict__ bd,
    int n_classes, float* __restrict__ probs)
{
    constexpr int H = 64, BUF = 10 * QL_STRIP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lim = count ? min(*count, n_rows) : n_rows;
    const int row_base = (int)blockIdx.x * 4;
    if (row_base >= lim) return;

    float4* tab = reinterpret_cast<float4*>(smem);
    unsigned char* hv = reinterpret_cast<unsigned char*>(smem + 4 * PXG_SIG_NSEG);     // [2 buffers][10 strips]
    int* ridx = reinterpret_cast<int*>(hv + 2 * BUF);

    const int tid = threadIdx.x, lane = tid & 63, W = tid >> 6;
    const int w4 = W >> 1, ntb = 2 * (W & 1);
    const int rd = lane & 3, slot = (lane >> 2) & 3, a = lane >> 4, kg = lane >> 4;

    for (int i = tid; i < PXG_SIG_NSEG; i += QL_THREADS) tab[i] = reinterpret_cast<const float4*>(sigtab)[i];
    v4i wq[2][8];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int d = 0; d < 8; d++) wq[i][d] = frag[((ntb + i) * 8 + d) * LSTM_THREADS + w4 * 64 + lane];
    for (int i = tid; i < 2 * BUF / 4; i += QL_THREADS) reinterpret_cast<unsigned*>(hv)[i] = 0u;
    if (tid < 4) {
        const int row = row_base + tid;
        ridx[tid] = row < lim ? (idx ? idx[row] : row) : -1;
    }
    // slot 3 carries this wave's first gate tile, slot 0 its second (ql_levels2)
    const bool publishes = slot == 3 || slot == 0;
    const int unit = 16 * w4 + 4 * (ntb + (slot == 0 ? 1 : 0)) + a;
    float stB[4];
#pragma unroll
    for (int r = 0; r < 4; r++) stB[r] = b3[r * H + unit];
    const int off0 = kg * QL_STRIP + (lane & 15) * 16;                                          // block 0: strips 0-3
    const int off1 = (4 + kg) * QL_STRIP + (lane & 15) * 16;                                    // block 1: strips 4-7
    const int off4 = (8 + (kg & 1)) * QL_STRIP + (2 - (kg >> 1)) * 64 + (lane & 15) * 16;      // half block: (w2 | w1)
    const int off5 = (8 + (kg & 1)) * QL_STRIP + (lane & 15) * 16;                             // half block: (w0 | 0)
    const int pub = (6 + (unit >> 4)) * QL_STRIP + rd * 16 + (unit & 15);
    const int sdir = tid / 36, spiece = tid % 36;
    const int s_lds = (sdir * 3 + spiece / 12) * QL_STRIP + 128 + (spiece % 12) * 16;
    const unsigned char* tile_in = bidir + (size_t)blockIdx.x * T * (2 * QL_SEQ);
    __syncthreads();
    if (tid < 72) *reinterpret_cast<f32x4*>(hv + s_lds) = *reinterpret_cast<const f32x4*>(tile_in + sdir * QL_SEQ + spiece * 16);
    __syncthreads();

    float C = 0.0f;
    float k65536;
    asm volatile("v_mov_b32 %0, 0x47800000" : "=v"(k65536));
    const v4i z = {0, 0, 0, 0};
    for (int t = 0; t < T; t++) {
        const unsigned char* hr = hv + (t & 1) * BUF;
        unsigned char* hw = hv + ((t + 1) & 1) * BUF;
        f32x4 pf = {0.f, 0.f, 0.f, 0.f};
        if (tid < 72 && t + 1 < T)
            pf = *reinterpret_cast<const f32x4*>(tile_in + ((size_t)(t + 1) * 2 + sdir) * QL_SEQ + spiece * 16);
        v4i acc[2];
        {
            const v4i f2 = *reinterpret_cast<const v4i*>(hr + off0 + 128);
            const v4i f1 = *reinterpret_cast<const v4i*>(hr + off0 + 64);
            const v4i f0 = *reinterpret_cast<const v4i*>(hr + off0);
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i] = mfma8(wq[i][0], f2, z);
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i] = mfma8(wq[i][1], f1, acc[i]);
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i] = mfma8(wq[i][2], f0, acc[i]);
        }
        {
            const v4i f2 = *reinterpret_cast<const v4i*>(hr + off1 + 128);
            const v4i f1 = *reinterpret_cast<const v4i*>(hr + off1 + 64);
            const v4i f0 = *reinterpret_cast<const v4i*>(hr + off1);
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i] = mfma8(wq[i][3], f2, acc[i]);
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i] = mfma8(wq[i][4], f1, acc[i]);
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i] = mfma8(wq[i][5], f0, acc[i]);
        }
        {
            const v4i f4 = *reinterpret_cast<const v4i*>(hr + off4);
            const v4i f5 = *reinterpret_cast<const v4i*>(hr + off5);
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i] = mfma8(wq[i][6], f4, acc[i]);
#pragma unroll
            for (int i = 0; i < 2; i++) acc[i] = mfma8(wq[i][7], f5, acc[i]);
        }
        f32x4 u[1];
#pragma unroll
        for (int r = 0; r < 4; r++) u[0][r] = __builtin_fmaf(ql_levels2(acc[0][r], acc[1][r], k65536), s3.g[r], stB[r]);
        float Cn[1] = {C}, hn[1];
        cells_update<1>(tab, u, Cn, hn);
        C = Cn[0];
        const unsigned q = q8_biased(hn[0]) ^ 0x00808080u;
        if (publishes) {
            hw[pub + 128] = (unsigned char)(q >> 16);
            hw[pub + 192] = (unsigned char)(q >> 8);
            hw[pub + 256] = (unsigned char)q;
        }
        if (tid < 72) *reinterpret_cast<f32x4*>(hw + s_lds) = pf;
        __syncthreads();
    }

    // ---- Dense(n_classes) + softmax on h = q * 2^-22 ---------------------------
    const unsigned char* hf = hv + (T & 1) * BUF;
    if (tid < 4) {
        const int row = tid;
        const int rdg = ridx[row];
        if (rdg >= 0) {
            float zz[PXG_MAX_CLASSES], e[PXG_MAX_CLASSES];
#pragma unroll
            for (int j = 0; j < PXG_MAX_CLASSES; j++) {
                zz[j] = -__builtin_inff();
                if (j < n_classes) {
                    float acc = bd[j];
                    for (int k = 0; k < H; k++) {
                        const unsigned char* p = hf + (6 + (k >> 4)) * QL_STRIP + row * 16 + (k & 15);
                        const int qq = (int)(signed char)p[256] + 256 * (int)(signed char)p[192] + 65536 * (int)(signed char)p[128];
                        acc = __builtin_fmaf((float)qq * (1.0f / 4194304.0f), Wd[k * n_classes + j], acc);
                    }
                    zz[j] = acc;
                }
            }
            float mx = zz[0];
#pragma unroll
            for (int j = 1; j < PXG_MAX_CLASSES; j++) mx = (j < n_classes && zz[j] > mx) ? zz[j] : mx;
            float sden = 0.0f;
#pragma unroll
            for (int j = 0; j < PXG_MAX_CLASSES; j++) {
                e[j] = j < n_classes ? pxg_expf(zz[j] - mx) : 0.0f;
                if (j < n_classes) sden = (j == 0) ? e[0] : sden + e[j];
            }
#pragma unroll
            for (int j = 0; j < PXG_MAX_CLASSES; j++)
                probs[(size_t)rdg * PXG_MAX_CLASSES + j] = j < n_classes ? e[j] / sden : 0.0f;
        }
    }
}

// ===========================================================================
// host side
// ===========================================================================
int pxg_launch_scaler_lstm_q8_lat(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx, const int32_t* count,
                                  const float* head, float* pred)
{
    if (n_rows <= 0) return PXG_OK;
    const int T = ctx->cfg.scaler_length / ctx->cfg.stride;
    const int64_t grid = (n_rows + 3) / 4;
    const size_t lds = sizeof(float) * 4 * PXG_SIG_NSEG + 2 * QL_HV + sizeof(float) * 2 * 4 * QL_XS + sizeof(int) * 4;
    const PxgLstmDev &l1 = ctx->scaler1, &l2 = ctx->scaler2;
    const Q8Scale s1 = q8_scale(ctx->q8.s_scaler1), s2 = q8_scale(ctx->q8.s_scaler2);
    // PXG_K2_LAT_WAVES=12: two gate tiles per wave on 12 waves -- shorter dependent chains per step, but more gate math on
    // half-used lanes: measured SLOWER (1 024 reads 1.59 -> 1.70 ms, 2 048 reads 2.33 -> 3.25: profiles/r06/ab_lat_forms.txt);
    // kept selectable, tested in both forms
    static const bool twelve = [] { const char* e = getenv("PXG_K2_LAT_WAVES"); return e && atoi(e) == 12; }();
    if (twelve)
        hipLaunchKernelGGL(k_scaler_lstm_q8_lat<2>, dim3((unsigned)grid), dim3(768), lds, ctx->stream, (int)n_rows, idx,
                           count, T, head, ctx->d_sigtab, reinterpret_cast<const v4i*>(ctx->q8.scaler_frag_lat), l1.kernel,
                           l1.bias, l2.bias, s1, s2, ctx->scaler_dense.kernel, ctx->scaler_dense.bias, pred);
    else
        hipLaunchKernelGGL(k_scaler_lstm_q8_lat<3>, dim3((unsigned)grid), dim3(512), lds, ctx->stream, (int)n_rows, idx,
                           count, T, head, ctx->d_sigtab, reinterpret_cast<const v4i*>(ctx->q8.scaler_frag_lat), l1.kernel,
                           l1.bias, l2.bias, s1, s2, ctx->scaler_dense.kernel, ctx->scaler_dense.bias, pred);
    PXG_HIP(ctx, hipGetLastError());
    return PXG_OK;
}

// bidir: ceil(n_rows / 4) * T * 2 * QL_SEQ bytes (below the tile form's ceil(n_rows / 16) * T * 2 * Q8_HVEC)
int pxg_launch_demux_lstm_q8_lat(pxg_ctx* ctx, int64_t n_rows, const int32_t* idx, const int32_t* count,
                                 const float* win, float* bidir, float* probs, int timer_a, int timer_b)
{
    if (n_rows <= 0) return PXG_OK;
    const int T = ctx->cfg.signal_trim_length;
    const int64_t grid = (n_rows + 3) / 4;
    const PxgLstmDev &f = ctx->demux_fwd, &b = ctx->demux_bwd, &t3 = ctx->demux_top;
    {
        const size_t lds = sizeof(float) * 4 * PXG_SIG_NSEG + 2 * 7 * QL_STRIP + sizeof(float) * 4 * T + sizeof(int) * 4;
        PXG_HIP(ctx, hipFuncSetAttribute((const void*)k_demux_bidir_q8_lat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        pxg_timer_begin(ctx, timer_a);
        hipLaunchKernelGGL(k_demux_bidir_q8_lat, dim3((unsigned)grid), dim3(QL_THREADS), lds, ctx->stream, (int)n_rows, idx,
                           count, T, win, ctx->d_sigtab, reinterpret_cast<const v4i*>(ctx->q8.bidir_frag_lat), f.kernel, f.bias,
                           b.kernel, b.bias, q8_scale(ctx->q8.s_fwd), q8_scale(ctx->q8.s_bwd), reinterpret_cast<unsigned char*>(bidir));
        pxg_timer_end(ctx, timer_a);
    }
    {
        const size_t lds = sizeof(float) * 4 * PXG_SIG_NSEG + 2 * 10 * QL_STRIP + sizeof(int) * 4;
        pxg_timer_begin(ctx, timer_b);
        hipLaunchKernelGGL(k_demux_top_q8_lat, dim3((unsigned)grid), dim3(QL_THREADS), lds, ctx->stream, (int)n_rows, idx, count,
                           T, reinterpret_cast<const unsigned char*>(bidir), ctx->d_sigtab,
                           reinterpret_cast<const v4i*>(ctx->q8.top_frag_lat), t3.bias, q8_scale(ctx->q8.s_top), ctx->demux_dense.kernel,
                           ctx->demux_dense.bias, ctx->demux_dense.out_dim, probs);
        pxg_timer_end(ctx, timer_b);
    }
    PXG_HIP(ctx, hipGetLastError());
    return PXG_OK;
}
