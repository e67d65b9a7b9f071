// Is v_fract_f32(x) == min(RN(x - floor(x)), 0x1.fffffep-1f) and v_cvt_flr_i32_f32(x) == (int)floor(x) for EVERY float
// with |x| <= 1024 on gfx950?  (The spline look-up of the LSTM kernels needs floor(u) as an index and u - floor(u) as
// the position inside the segment: v_floor + v_sub + v_cvt today, v_fract + v_cvt_flr if this holds -- with the
// clamp below 1 written into the arithmetic's definition, oracle/pxo_core.c sig_lookup_u.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k(unsigned long long* bad_fract, unsigned long long* bad_flr, unsigned long long* clamped, unsigned* first)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bf = 0, bi = 0, cl = 0;
    for (uint64_t b = tid; b < (1ull << 32); b += step) {
        const float x = __uint_as_float((unsigned)b);
        if (!(__builtin_fabsf(x) <= 1024.0f)) continue;            // also skips NaN
        const float fl = __builtin_floorf(x);
        float ref = x - fl;                                         // v_floor_f32 + v_sub_f32: what the kernels do today
        if (ref >= 1.0f) { ref = 0.99999994f; cl++; }
        const float hw = __builtin_amdgcn_fractf(x);
        int hi;
        asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(hi) : "v"(x));
        if (__float_as_uint(hw) != __float_as_uint(ref)) {
            if (!(hw == 0.0f && ref == 0.0f)) {                     // (+0 vs -0 would not matter to the spline: count separately below)
                bf++;
                atomicMin(first, (unsigned)b);
            }
        }
        if (hi != (int)fl) bi++;
    }
    atomicAdd(bad_fract, bf);
    atomicAdd(bad_flr, bi);
    atomicAdd(clamped, cl);
}

int main()
{
    unsigned long long *d, h[3] = {0, 0, 0};
    unsigned* first; unsigned hf = 0xFFFFFFFFu;
    hipMalloc(&d, 3 * sizeof(*d)); hipMalloc(&first, 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(first, &hf, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, d + 1, d + 2, first);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("every float with |x| <= 1024: v_fract_f32 != min(x - floor(x), 0x1.fffffep-1f) for %llu values (first bit pattern %08x); "
           "v_cvt_flr_i32_f32 != (int)floor(x) for %llu; x - floor(x) rounds to 1.0 (clamped) for %llu values\n", h[0], hf, h[1], h[2]);
    return 0;
}
