// Issue rate of the K = 32 int8 MFMA (v_mfma_i32_16x16x32_i8, the CDNA3 shape) next to gfx950's K = 64 one:
// would a 96-wide layer (two 48-unit inputs) be cheaper as three K = 32 blocks than as two K = 64 blocks?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int WAVES, int SHAPE>
__global__ __launch_bounds__(WAVES * 64) void k_rate(int* out, int iters)
{
    v4i acc[9];
    for (int i = 0; i < 9; i++) acc[i] = v4i{0, 0, 0, 0};
    v4i a = {(int)threadIdx.x, 3, 5, 7}, b = {1, (int)threadIdx.x * 3, 9, 11};
    long a8 = threadIdx.x * 0x0101010101010101L, b8 = 0x0302010003020100L + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 54; i++) {
            if (SHAPE == 64) acc[i % 9] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i % 9], 0, 0, 0);
            else acc[i % 9] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, acc[i % 9], 0, 0, 0);
        }
    }
    int s = 0;
    for (int i = 0; i < 9; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WAVES, int SHAPE>
void run(const char* name)
{
    int* out;
    const int blocks = 256, iters = 2000;
    hipMalloc(&out, blocks * WAVES * 64 * sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<WAVES, SHAPE>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<WAVES, SHAPE>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // one workgroup per CU: WAVES / 4 waves per SIMD
    const double per_simd = (double)iters * 54 * (WAVES / 4.0);
    printf("%-40s %d wave(s)/SIMD: %.3f ms  %.1f cyc per MFMA per SIMD at 2.4 GHz\n", name, WAVES / 4, ms, ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(out);
}

int main()
{
    run<4, 64>("v_mfma_i32_16x16x64_i8");
    run<8, 64>("v_mfma_i32_16x16x64_i8");
    run<4, 32>("v_mfma_i32_16x16x32_i8");
    run<8, 32>("v_mfma_i32_16x16x32_i8");
    return 0;
}
