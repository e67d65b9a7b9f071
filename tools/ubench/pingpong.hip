// Can a SIMD run one wave's int8 MFMAs under ANOTHER wave's VALU work, and does it take a barrier-
// enforced anti-phase ("ping-pong") to get it?  The K2/K5 step of k_lstm_q8.hip is a MFMA phase
// (72 v_mfma_i32_16x16x64_i8 back to back) followed by a VALU phase (~470 instructions, half of them
// conversions / shifts / med3); two such waves share a SIMD.  Modes, 8 waves = 2 per SIMD:
//   0  free running, every wave: MFMA phase then VALU phase (what two 4-wave workgroups per CU do)
//   1  barrier after each phase, both halves in the SAME phase (worst case of free running)
//   2  barrier after each phase, halves in OPPOSITE phases (ping-pong)
//   3  MFMA phase only      4  VALU phase only
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

#define NMF 72
// 16 VALU: 8 plain f32 (fma / add / mul) + 8 "slow" (cvt, lshl_add, med3, floor), independent chains
#define VMIX(a0, a1, a2, a3, a4, a5, a6, a7, i0, i1, i2, i3)                                                  \
    asm volatile("v_fma_f32 %0, %0, %12, %13\n v_cvt_f32_i32 %4, %8\n v_fmac_f32 %1, %12, %13\n v_lshl_add_u32 %9, %8, 8, %9\n"   \
                 "v_add_f32 %2, %2, %12\n v_med3_f32 %5, %5, %12, %13\n v_mul_f32 %3, %3, %13\n v_floor_f32 %6, %6\n"            \
                 "v_fma_f32 %0, %0, %13, %12\n v_cvt_i32_f32 %10, %7\n v_fmac_f32 %1, %13, %12\n v_lshl_add_u32 %11, %10, 4, %11\n" \
                 "v_add_f32 %2, %2, %13\n v_med3_f32 %5, %5, %13, %12\n v_mul_f32 %3, %3, %12\n v_cvt_f32_i32 %7, %9\n"            \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) \
                 : "v"(c0), "v"(c1));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters)
{
    const int half = threadIdx.x >> 8;
    v4i acc[12];
    for (int i = 0; i < 12; i++) acc[i] = v4i{0, 0, 0, 0};
    v4i a = {(int)threadIdx.x, 3, 5, 7}, b = {1, (int)threadIdx.x * 3, 9, 11};
    float c0 = 1.0001f, c1 = 0.9999f;
    float f0 = threadIdx.x, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
    float g0 = threadIdx.x, g1 = 1, g2 = 2, g3 = 3, g4 = 4, g5 = 5, g6 = 6, g7 = 7;
    int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3, j0 = 5, j1 = 6, j2 = 7, j3 = 8;
    auto mfma_phase = [&]() {
#pragma unroll
        for (int i = 0; i < NMF; i++) acc[i % 12] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i % 12], 0, 0, 0);
    };
    auto valu_phase = [&]() {
#pragma unroll
        for (int i = 0; i < 15; i++) {          // 15 x 32 = 480 VALU
            VMIX(f0, f1, f2, f3, f4, f5, f6, f7, i0, i1, i2, i3)
            VMIX(g0, g1, g2, g3, g4, g5, g6, g7, j0, j1, j2, j3)
        }
    };
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { mfma_phase(); valu_phase(); }
        if (MODE == 1) { mfma_phase(); __builtin_amdgcn_s_barrier(); valu_phase(); __builtin_amdgcn_s_barrier(); }
        if (MODE == 2) {
            if (half == 0) mfma_phase(); else valu_phase();
            __builtin_amdgcn_s_barrier();
            if (half == 0) valu_phase(); else mfma_phase();
            __builtin_amdgcn_s_barrier();
        }
        if (MODE == 3) mfma_phase();
        if (MODE == 4) valu_phase();
    }
    int s = 0;
    for (int i = 0; i < 12; i++) s += acc[i][0] + acc[i][2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(s + i0 + i1 + i2 + i3 + j0 + j1 + j2 + j3) + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7;
}

template <int MODE>
static float run(float* out, int iters)
{
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 10);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 3000;
    const char* names[] = {"free running (MFMA phase, VALU phase)", "barriers, both waves of a SIMD in the SAME phase",
                           "barriers, OPPOSITE phases (ping-pong)", "MFMA phase only (2 waves/SIMD)", "VALU phase only (2 waves/SIMD)"};
    float ms[5] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters)};
    for (int m = 0; m < 5; m++)
        printf("%-52s %.3f ms  %.0f cycles per (72 MFMA + 480 VALU) of ONE wave at 2.4 GHz\n", names[m], ms[m],
               ms[m] * 1e-3 * 2.4e9 / iters / (m >= 3 ? 1 : 1));
    printf("(every figure is the time for BOTH waves of a SIMD to finish one iteration each)\n");
    return 0;
}
