// v_mfma_i32_32x32x32_i8 on gfx950: is the 32-cycle shape a better carrier for the
// exact-integer LSTM products than v_mfma_i32_16x16x64_i8 (tools/ubench/i8_mfma.hip)?
//   (a) exactness: the exact integer dot product when A and B use the same (lane half, byte) -> k map;
//       the C/D layout (row = 8 (r >> 2) + 4 (lane >> 5) + (r & 3), column = lane & 31) is checked too;
//   (b) issue cycles per instruction with 1 and 2 waves per SIMD, independent and dependent accumulators;
//   (c) how many INDEPENDENT single-issue instructions (plain f32 VALU, "slow" VALU = cvt / lshl_add / med3,
//       ds_read_b128) a wave -- and the other wave of its SIMD -- issue inside the instruction's shadow;
//   (d) the same harness on the 16x16x64 shape with HALF the fillers per MFMA (equal work per filler).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k_check(const int8_t* A /*[32][32]*/, const int8_t* Bt /*[32][32]: column j's k vector*/, int* C /*[32][32]*/)
{
    const int lane = threadIdx.x, rc = lane & 31, g = lane >> 5;
    v4i a, b;
    const int* pa = reinterpret_cast<const int*>(A + rc * 32 + g * 16);
    const int* pb = reinterpret_cast<const int*>(Bt + rc * 32 + g * 16);
    for (int i = 0; i < 4; i++) { a[i] = pa[i]; b[i] = pb[i]; }
    v16i c;
    for (int i = 0; i < 16; i++) c[i] = 0;
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) C[(8 * (r >> 2) + 4 * g + (r & 3)) * 32 + rc] = c[r];
}

// fillers: NF instructions after every MFMA.  KIND 0: v_fma_f32 (8 independent chains); 1: the combine-step mix
// (v_cvt_f32_i32, v_lshl_add_u32, v_med3_f32, v_fma_f32); 2: ds_read_b128 (conflict-free, results summed at the end)
#define F1(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(bf));
#define S_CVT(d, s) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(d) : "v"(s));
#define S_LSHL(d, s) asm volatile("v_lshl_add_u32 %0, %1, 8, %0" : "+v"(d) : "v"(s));
#define S_MED(d) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(d) : "v"(bf));

template <int KIND, int NF>
__device__ __forceinline__ void fillers(float (&v)[8], int (&iv)[4], float bf, const v4i* lds, v4i (&sink)[2], int lane)
{
    if constexpr (KIND == 0) {
#pragma unroll
        for (int i = 0; i < NF; i++) F1(v[i & 7])
    } else if constexpr (KIND == 1) {
#pragma unroll
        for (int i = 0; i < NF; i++) {
            if ((i & 3) == 0) S_CVT(v[i & 7], iv[0])
            if ((i & 3) == 1) S_LSHL(iv[1], iv[2])
            if ((i & 3) == 2) S_MED(v[(i + 3) & 7])
            if ((i & 3) == 3) F1(v[(i + 5) & 7])
        }
    } else if constexpr (KIND == 3) {          // ONE dependent chain (the combine of a gate: each instruction needs the last)
#pragma unroll
        for (int i = 0; i < NF; i++) F1(v[0])
    } else {
#pragma unroll
        for (int i = 0; i < NF; i++) {
            v4i t;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(lane * 16), "n"((i & 7) * 1024));
            sink[i & 1] = t;          // (waits are the compiler's: the value is only used after the loop)
        }
    }
}

// ACC 0: the builtin (hipcc keeps these accumulators in AGPRs); 1: inline asm, accumulators forced into VGPRs ("+v": what a
// kernel whose VALU reads the results gets); 2: inline asm, AGPRs ("+a")
template <int ACC>
__device__ __forceinline__ void mf32(v16i& c, v4i a, v4i b)
{
    if constexpr (ACC == 0) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    else if constexpr (ACC == 1) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int ACC>
__device__ __forceinline__ void mf16(v4i& c, v4i a, v4i b)
{
    if constexpr (ACC == 0) c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    else if constexpr (ACC == 1) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// SHAPE 32: 32x32x32 (NM per iteration, 4 independent accumulators of 16 registers, or ONE when DEP);
// SHAPE 16: 16x16x64 (2 NM per iteration, 8 independent accumulators of 4 registers), NF / 2 fillers after each
template <int WAVES, int SHAPE, int KIND, int NF, bool DEP, int MFMA_WAVES /* waves [0, n) issue MFMAs, the rest only fillers; 0 = all both */, int ACC = 0>
__global__ __launch_bounds__(WAVES * 64) void k_rate(float* out, int iters)
{
    __shared__ v4i lds[8 * 64 + 64];
    constexpr int NM = 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8 * 64 + 64; i += WAVES * 64) lds[i] = v4i{i, 1, 2, 3};
    __syncthreads();
    v16i acc32[4];
    v4i acc16[8];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) acc32[i][j] = 0;
    for (int i = 0; i < 8; i++) acc16[i] = v4i{0, 0, 0, 0};
    v4i a = {(int)threadIdx.x, 3, 5, 7}, b = {1, (int)threadIdx.x * 3, 9, 11};
    float bf = 1.0001f;
    float v[8];
    int iv[4] = {(int)threadIdx.x, 3, 77, 5};
    v4i sink[2] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 1e-3f + i;
    const bool do_m = MFMA_WAVES == 0 || wave < MFMA_WAVES;
    const bool do_f = MFMA_WAVES == 0 || wave >= MFMA_WAVES;
    for (int it = 0; it < iters; it++) {
        if (MFMA_WAVES == 0) {
#pragma unroll
            for (int i = 0; i < NM; i++) {
                if constexpr (SHAPE == 32) {
                    mf32<ACC>(acc32[DEP ? 0 : (i & 3)], a, b);
                    __builtin_amdgcn_sched_barrier(0);
                    fillers<KIND, NF>(v, iv, bf, lds, sink, lane);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    mf16<ACC>(acc16[DEP ? 0 : ((2 * i) & 7)], a, b);
                    __builtin_amdgcn_sched_barrier(0);
                    fillers<KIND, (NF + 1) / 2>(v, iv, bf, lds, sink, lane);
                    __builtin_amdgcn_sched_barrier(0);
                    mf16<ACC>(acc16[DEP ? 0 : ((2 * i + 1) & 7)], a, b);
                    __builtin_amdgcn_sched_barrier(0);
                    fillers<KIND, NF / 2>(v, iv, bf, lds, sink, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int i = 0; i < NM; i++) {
                    if constexpr (SHAPE == 32) acc32[i & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc32[i & 3], 0, 0, 0);
                    else {
                        acc16[(2 * i) & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc16[(2 * i) & 7], 0, 0, 0);
                        acc16[(2 * i + 1) & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc16[(2 * i + 1) & 7], 0, 0, 0);
                    }
                }
            }
            if (do_f) {
#pragma unroll
                for (int i = 0; i < NM; i++) fillers<KIND, NF>(v, iv, bf, lds, sink, lane);
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 4; i++) s += acc32[i][0] + acc32[i][15] + acc32[i][7];
    for (int i = 0; i < 8; i++) s += acc16[i][0] + acc16[i][3];
    float f = 0;
    for (int i = 0; i < 8; i++) f += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(s + iv[0] + iv[1] + sink[0][0] + sink[1][3]) + f;
}

static const double CLK = 2.4e9;

template <int WAVES, int SHAPE, int KIND, int NF, bool DEP, int MW, int ACC = 0>
static float run(float* out, int iters)
{
    hipLaunchKernelGGL((k_rate<WAVES, SHAPE, KIND, NF, DEP, MW, ACC>), dim3(256), dim3(WAVES * 64), 0, 0, out, 10);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<WAVES, SHAPE, KIND, NF, DEP, MW, ACC>), dim3(256), dim3(WAVES * 64), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

// cycles per 32x32x32-equivalent (= two 16x16x64) per SIMD
template <int SHAPE, int KIND, int NF>
static void line(float* out, int iters, const char* what)
{
    const float m1 = run<4, SHAPE, KIND, NF, false, 0>(out, iters), m2 = run<8, SHAPE, KIND, NF, false, 0>(out, iters);
    printf("%-9s %2d %-10s after each 32x32x32-equivalent:  1 wave/SIMD %6.1f cyc   2 waves/SIMD %6.1f cyc per SIMD\n",
           SHAPE == 32 ? "32x32x32" : "16x16x64", NF, what, m1 * 1e-3 * CLK / (iters * 32.0), m2 * 1e-3 * CLK / (iters * 64.0));
}

template <int SHAPE, int ACC, int NF>
static void accline(float* out, int iters)
{
    const float m1 = run<4, SHAPE, 0, NF, false, 0, ACC>(out, iters), m2 = run<8, SHAPE, 0, NF, false, 0, ACC>(out, iters);
    const float x1 = run<4, SHAPE, 1, NF, false, 0, ACC>(out, iters), x2 = run<8, SHAPE, 1, NF, false, 0, ACC>(out, iters);
    printf("%-9s accumulators in %s (inline asm), %2d fillers per 32x32x32-equivalent: v_fma_f32 %6.1f / %6.1f cyc, mixed VALU %6.1f / %6.1f (1 / 2 waves per SIMD)\n",
           SHAPE == 32 ? "32x32x32" : "16x16x64", ACC == 1 ? "VGPRs" : "AGPRs", NF, m1 * 1e-3 * CLK / (iters * 32.0), m2 * 1e-3 * CLK / (iters * 64.0),
           x1 * 1e-3 * CLK / (iters * 32.0), x2 * 1e-3 * CLK / (iters * 64.0));
}

template <int SHAPE, int KIND>
static void sweep(float* out, int iters, const char* what)
{
    line<SHAPE, KIND, 0>(out, iters, what);
    line<SHAPE, KIND, 2>(out, iters, what);
    line<SHAPE, KIND, 4>(out, iters, what);
    line<SHAPE, KIND, 6>(out, iters, what);
    line<SHAPE, KIND, 8>(out, iters, what);
    line<SHAPE, KIND, 12>(out, iters, what);
    line<SHAPE, KIND, 16>(out, iters, what);
}

int main()
{
    // ---- (a) exactness / layout --------------------------------------------------------
    std::vector<int8_t> A(32 * 32), Bt(32 * 32);
    srand(922);
    for (auto& x : A) x = (int8_t)(rand() % 256 - 128);
    for (auto& x : Bt) x = (int8_t)(rand() % 256 - 128);
    int8_t *dA, *dB; int* dC;
    (void)hipMalloc(&dA, 1024); (void)hipMalloc(&dB, 1024); (void)hipMalloc(&dC, 4096);
    (void)hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<int> C(1024);
    (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            int s = 0;
            for (int k = 0; k < 32; k++) s += (int)A[i * 32 + k] * (int)Bt[j * 32 + k];
            if (s != C[i * 32 + j]) bad++;
        }
    printf("exactness: %d of 1024 outputs differ from the integer dot product (k = 16 (lane >> 5) + byte for A and B; "
           "D row = 8 (r >> 2) + 4 (lane >> 5) + (r & 3), column = lane & 31)\n", bad);

    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 3000;
    // ---- (b) issue rate ----------------------------------------------------------------
    {
        const float a1 = run<4, 32, 0, 0, false, 0>(out, iters), a2 = run<8, 32, 0, 0, false, 0>(out, iters);
        const float d1 = run<4, 32, 0, 0, true, 0>(out, iters), d2 = run<8, 32, 0, 0, true, 0>(out, iters);
        const float b1 = run<4, 16, 0, 0, false, 0>(out, iters), b2 = run<8, 16, 0, 0, false, 0>(out, iters);
        printf("32x32x32 i8, 4 independent accumulators: 1 wave/SIMD %.1f cyc per MFMA, 2 waves/SIMD %.1f per SIMD (at 2.4 GHz)\n",
               a1 * 1e-3 * CLK / (iters * 32.0), a2 * 1e-3 * CLK / (iters * 64.0));
        printf("32x32x32 i8, ONE accumulator (dependent):  1 wave/SIMD %.1f cyc per MFMA, 2 waves/SIMD %.1f per SIMD\n",
               d1 * 1e-3 * CLK / (iters * 32.0), d2 * 1e-3 * CLK / (iters * 64.0));
        printf("16x16x64 i8, 8 independent accumulators: 1 wave/SIMD %.1f cyc per PAIR, 2 waves/SIMD %.1f per SIMD\n",
               b1 * 1e-3 * CLK / (iters * 32.0), b2 * 1e-3 * CLK / (iters * 64.0));
    }
    // ---- (c), (d) fillers in the shadow ------------------------------------------------
    sweep<32, 0>(out, iters, "v_fma_f32");
    sweep<16, 0>(out, iters, "v_fma_f32");
    sweep<32, 1>(out, iters, "mixed VALU");
    sweep<16, 1>(out, iters, "mixed VALU");
    sweep<32, 2>(out, iters, "ds_read128");
    sweep<16, 2>(out, iters, "ds_read128");
    // ---- where the accumulators live: AGPRs (the lines above) or VGPRs (a kernel whose VALU reads them) ----
    accline<32, 1, 0>(out, iters); accline<32, 1, 4>(out, iters); accline<32, 1, 8>(out, iters); accline<32, 1, 16>(out, iters);
    accline<16, 1, 0>(out, iters); accline<16, 1, 4>(out, iters); accline<16, 1, 8>(out, iters); accline<16, 1, 16>(out, iters);
    accline<32, 2, 0>(out, iters); accline<32, 2, 4>(out, iters); accline<32, 2, 8>(out, iters); accline<32, 2, 16>(out, iters);
    accline<16, 2, 0>(out, iters); accline<16, 2, 4>(out, iters); accline<16, 2, 8>(out, iters); accline<16, 2, 16>(out, iters);
    {
        const float d4a = run<4, 16, 3, 4, false, 0, 2>(out, iters), d4b = run<8, 16, 3, 4, false, 0, 2>(out, iters);
        const float d8a = run<4, 16, 3, 8, false, 0, 2>(out, iters), d8b = run<8, 16, 3, 8, false, 0, 2>(out, iters);
        printf("16x16x64, AGPR accumulators, fillers = ONE dependent v_fma_f32 chain: 4 per pair %.1f / %.1f cyc (1 / 2 waves per SIMD), 8 per pair %.1f / %.1f\n",
               d4a * 1e-3 * CLK / (iters * 32.0), d4b * 1e-3 * CLK / (iters * 64.0), d8a * 1e-3 * CLK / (iters * 32.0), d8b * 1e-3 * CLK / (iters * 64.0));
    }
    // ---- MFMA wave + filler wave on one SIMD ---------------------------------------------
    {
        const float f8 = run<4, 32, 0, 8, false, -1>(out, iters);                // the filler wave alone (1 per SIMD)
        printf("one wave per SIMD issuing 256 v_fma_f32 per iteration, alone: %.3f ms\n", f8);
        const float m = run<4, 32, 0, 0, false, 4>(out, iters);                  // 1 MFMA wave per SIMD alone
        const float v = run<8, 32, 0, 8, false, 4>(out, iters);                  // + 1 wave per SIMD issuing 8 fma per "slot"
        const float v16 = run<8, 16, 0, 8, false, 4>(out, iters);
        const float m16 = run<4, 16, 0, 0, false, 4>(out, iters);
        printf("one MFMA wave per SIMD (32 x 32x32x32 per iteration) alone %.3f ms; with a second wave issuing 256 v_fma_f32 per iteration %.3f ms\n", m, v);
        printf("one MFMA wave per SIMD (64 x 16x16x64 per iteration) alone %.3f ms; with a second wave issuing 256 v_fma_f32 per iteration %.3f ms\n", m16, v16);
    }
    return bad != 0;
}
