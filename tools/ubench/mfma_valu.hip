// Do v_mfma_f32_16x16x4_f32 and plain f32 VALU overlap on one SIMD?
// blocks of 512 threads = 2 waves per SIMD; mode 0: all waves MFMA; 1: all waves VALU;
// 2: waves 0-3 MFMA + waves 4-7 VALU (one of each per SIMD); 3: every wave interleaves both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NM 64     // MFMAs per iteration
#define NV 512    // VALU per iteration
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode)
{
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    const bool do_m = mode == 0 || (mode == 2 && wave < 4) || mode == 3;
    const bool do_v = mode == 1 || (mode == 2 && wave >= 4) || mode == 3;
    for (int it = 0; it < iters; it++) {
        if (mode == 3) {
#pragma unroll
            for (int i = 0; i < NM; i++) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\n"
                             "v_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b));
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int i = 0; i < NM; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int i = 0; i < NV / 8; i++)
                    asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\n"
                                 "v_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8\n"
                                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
int main()
{
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 2000;
    const char* names[] = {"all waves MFMA (2/SIMD)", "all waves VALU (2/SIMD)", "1 MFMA wave + 1 VALU wave per SIMD", "each wave interleaves 1 MFMA : 8 VALU"};
    for (int mode = 0; mode < 4; mode++) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 10, mode);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        double n_m = (mode == 0 || mode == 3) ? 2.0 * NM : (mode == 2 ? NM : 0);     // MFMAs per SIMD per iter
        double n_v = (mode == 1) ? 2.0 * NV : (mode == 2 ? NV : (mode == 3 ? 2.0 * NM * 8 : 0));
        printf("%-40s %.3f ms   per SIMD-iter: %.0f MFMA + %.0f VALU -> %.1f ns/iter\n", names[mode], ms, n_m, n_v, ms * 1e6 / iters);
    }
}
