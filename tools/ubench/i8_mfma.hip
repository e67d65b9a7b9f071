// v_mfma_i32_16x16x64_i8 on gfx950: (1) is the result the exact integer dot product when A and B
// fragments use the same (lane group, byte) -> k map (the only layout fact the exact-integer LSTM
// arithmetic needs); (2) issue rate alone; (3) does it overlap with VALU work of ANOTHER wave on
// the same SIMD, and of the SAME wave (the f32 MFMA does neither: mfma_valu.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_check(const int8_t* A /*[16][64]*/, const int8_t* B /*[64][16] as Bt[16][64]*/, int* C /*[16][16]*/)
{
    const int lane = threadIdx.x, rc = lane & 15, g = lane >> 4;
    v4i a, b;
    const int* pa = reinterpret_cast<const int*>(A + rc * 64 + g * 16);
    const int* pb = reinterpret_cast<const int*>(B + rc * 64 + g * 16);
    for (int i = 0; i < 4; i++) { a[i] = pa[i]; b[i] = pb[i]; }
    v4i c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) C[(g * 4 + r) * 16 + rc] = c[r];      // row = 4*(lane>>4)+r, col = lane&15
}

#define NM 54
#define VBLK                                                                                          \
    asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\n" \
                 "v_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8\n" \
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(bf));
// integer-class VALU (the conversions / shifts of the combine step)
#define IBLK                                                                                          \
    asm volatile("v_cvt_f32_i32 %0, %4\nv_lshl_add_u32 %1, %5, 8, %6\nv_cvt_f32_i32 %2, %7\nv_med3_f32 %3, %3, %8, %8\n" \
                 : "+v"(v0), "+v"(i1), "+v"(v2), "+v"(v3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(bf));

// MODE 0: all waves MFMA; 1: all waves VALU; 2: waves [0, 4) MFMA, the rest VALU; 3: every wave NV VALU after each MFMA
template <int WAVES, int MODE, int NV>
__global__ __launch_bounds__(WAVES * 64) void k_rate(float* out, int iters)
{
    constexpr int mode = MODE, nv = NV;
    const int wave = threadIdx.x >> 6;
    v4i acc[9];
    for (int i = 0; i < 9; i++) acc[i] = v4i{0, 0, 0, 0};
    v4i a = {(int)threadIdx.x, 3, 5, 7}, b = {1, (int)threadIdx.x * 3, 9, 11};
    float af = threadIdx.x * 1e-3f, bf = 1.0001f;
    float v0 = af, v1 = af + 1, v2 = af + 2, v3 = af + 3, v4 = af + 4, v5 = af + 5, v6 = af + 6, v7 = af + 7;
    int i0 = threadIdx.x, i1 = 3, i2 = 77, i3 = 5;
    const bool do_m = mode == 0 || (mode == 2 && wave < 4);
    const bool do_v = mode == 1 || (mode == 2 && wave >= 4);
    for (int it = 0; it < iters; it++) {
        if (mode == 3) {          // same wave: nv VALU after every MFMA
#pragma unroll
            for (int i = 0; i < NM; i++) {
                acc[i % 9] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i % 9], 0, 0, 0);
                if constexpr (nv >= 8) { VBLK }
                if constexpr (nv >= 16) { VBLK }
                if constexpr (nv == 4 || nv == 12) { IBLK }
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int i = 0; i < NM; i++) acc[i % 9] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i % 9], 0, 0, 0);
            }
            if (do_v) {
                #pragma unroll
                for (int i = 0; i < nv / 8; i++) { VBLK }
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 9; i++) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)i1;
}

template <int WAVES, int MODE, int NV>
static float run(float* out, int iters)
{
    hipLaunchKernelGGL((k_rate<WAVES, MODE, NV>), dim3(256), dim3(WAVES * 64), 0, 0, out, 10);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<WAVES, MODE, NV>), dim3(256), dim3(WAVES * 64), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int NV>
static void same_wave(float* out, int iters, double clk)
{
    float ms = run<4, 3, NV>(out, iters);
    printf("same wave, %2d VALU after every MFMA, 1 wave/SIMD:  %.3f ms  %.1f cyc per group\n", NV, ms, ms * 1e-3 * clk / (iters * (double)NM));
    ms = run<8, 3, NV>(out, iters);
    printf("same wave, %2d VALU after every MFMA, 2 waves/SIMD: %.3f ms  %.1f cyc per group per SIMD\n", NV, ms, ms * 1e-3 * clk / (iters * 2.0 * NM));
}

int main()
{
    // ---- (1) exactness / layout --------------------------------------------------------
    std::vector<int8_t> A(16 * 64), Bt(16 * 64);
    srand(922);
    for (auto& x : A) x = (int8_t)(rand() % 256 - 128);
    for (auto& x : Bt) x = (int8_t)(rand() % 256 - 128);
    int8_t *dA, *dB; int* dC;
    (void)hipMalloc(&dA, 1024); (void)hipMalloc(&dB, 1024); (void)hipMalloc(&dC, 1024);
    (void)hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<int> C(256);
    (void)hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            int s = 0;
            for (int k = 0; k < 64; k++) s += (int)A[i * 64 + k] * (int)Bt[j * 64 + k];
            if (s != C[i * 16 + j]) bad++;
        }
    printf("exactness: %d of 256 outputs differ from the integer dot product (A/B loaded with the same (group, byte) -> k map)\n", bad);

    // ---- (2), (3) rates ----------------------------------------------------------------
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 4000;
    const double clk = 2.4e9;
    float ms;
    ms = run<4, 0, 0>(out, iters);
    printf("1 wave/SIMD, MFMA only:   %.3f ms  %.1f cyc/MFMA at 2.4 GHz\n", ms, ms * 1e-3 * clk / (iters * (double)NM));
    ms = run<8, 0, 0>(out, iters);
    printf("2 waves/SIMD, MFMA only:  %.3f ms  %.1f cyc/MFMA per SIMD\n", ms, ms * 1e-3 * clk / (iters * 2.0 * NM));
    const float v1 = run<4, 1, 432>(out, iters), v2 = run<8, 1, 432>(out, iters), v3 = run<12, 1, 432>(out, iters);
    printf("VALU only, 432 v_fma_f32 per wave-iteration: 1 / 2 / 3 waves per SIMD: %.3f / %.3f / %.3f ms (%.2f / %.2f / %.2f cyc per fma per SIMD)\n",
           v1, v2, v3, v1 * 1e-3 * clk / (iters * 432.0), v2 * 1e-3 * clk / (iters * 864.0), v3 * 1e-3 * clk / (iters * 1296.0));
    const float m2 = run<8, 2, 432>(out, iters), m3 = run<12, 2, 432>(out, iters);
    printf("1 MFMA wave (54 per iteration) + 1 VALU wave per SIMD: %.3f ms; + 2 VALU waves per SIMD: %.3f ms (no sharing: the VALU-only times above)\n", m2, m3);
    same_wave<0>(out, iters, clk);
    same_wave<4>(out, iters, clk);
    same_wave<8>(out, iters, clk);
    same_wave<12>(out, iters, clk);
    same_wave<16>(out, iters, clk);
    return bad != 0;
}
