// Micro-benchmark: issue cost (cycles per wave-instruction) of the VALU ops the
// LSTM gate math uses, 1 and 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP 256
template <int OP>
__global__ void k(float* out, int iters, long long* cyc)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    const unsigned long long msk = 0x5555555555555555ull;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (OP == 0) { // v_fma_f32, 8 independent chains
                asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (OP == 1) { // v_pk_fma_f32 on register pairs
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(*(f32x2*)&a0), "+v"(*(f32x2*)&a2), "+v"(*(f32x2*)&a4), "+v"(*(f32x2*)&a6)
                             : "v"(f32x2{b, b}), "v"(f32x2{c, c}));
            } else if (OP == 2) { // v_cndmask
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            } else if (OP == 3) { // dpp quad_perm mov
                asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 4) { // v_rcp_f32
                asm volatile("v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\nv_rcp_f32 %4, %4\nv_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 5) { // v_add_f32
                asm volatile("v_add_f32 %0, %0, %8\nv_add_f32 %1, %1, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\nv_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_add_f32 %6, %6, %8\nv_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            } else if (OP == 6) { // v_cvt_i32_f32 + v_lshl_add_u32 pair
                asm volatile("v_cvt_i32_f32 %0, %0\nv_lshl_add_u32 %1, %0, 23, %1\nv_cvt_i32_f32 %2, %2\nv_lshl_add_u32 %3, %2, 23, %3\nv_cvt_i32_f32 %4, %4\nv_lshl_add_u32 %5, %4, 23, %5\nv_cvt_i32_f32 %6, %6\nv_lshl_add_u32 %7, %6, 23, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 7) { // v_fma_f64
                double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
                asm volatile("v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0001), "v"(0.5));
                a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
            } else if (OP == 9) { // v_cndmask_b32_e64 with an SGPR-pair mask
                asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\nv_cndmask_b32_e64 %1, %1, %8, %9\nv_cndmask_b32_e64 %2, %2, %8, %9\nv_cndmask_b32_e64 %3, %3, %8, %9\n"
                             "v_cndmask_b32_e64 %4, %4, %8, %9\nv_cndmask_b32_e64 %5, %5, %8, %9\nv_cndmask_b32_e64 %6, %6, %8, %9\nv_cndmask_b32_e64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(msk));
            } else if (OP == 10) { // v_fmac_f32 (VOP2)
                asm volatile("v_fmac_f32 %0, %8, %9\nv_fmac_f32 %1, %8, %9\nv_fmac_f32 %2, %8, %9\nv_fmac_f32 %3, %8, %9\nv_fmac_f32 %4, %8, %9\nv_fmac_f32 %5, %8, %9\nv_fmac_f32 %6, %8, %9\nv_fmac_f32 %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (OP == 11) { // dependent v_fma chain (latency)
                asm volatile("v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(b), "v"(c));
            } else if (OP == 12) { // v_fract_f32
                asm volatile("v_fract_f32 %0, %0\nv_fract_f32 %1, %1\nv_fract_f32 %2, %2\nv_fract_f32 %3, %3\nv_fract_f32 %4, %4\nv_fract_f32 %5, %5\nv_fract_f32 %6, %6\nv_fract_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 13) { // v_cvt_flr_i32_f32
                asm volatile("v_cvt_flr_i32_f32 %0, %0\nv_cvt_flr_i32_f32 %1, %1\nv_cvt_flr_i32_f32 %2, %2\nv_cvt_flr_i32_f32 %3, %3\nv_cvt_flr_i32_f32 %4, %4\nv_cvt_flr_i32_f32 %5, %5\nv_cvt_flr_i32_f32 %6, %6\nv_cvt_flr_i32_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 14) { // v_floor_f32
                asm volatile("v_floor_f32 %0, %0\nv_floor_f32 %1, %1\nv_floor_f32 %2, %2\nv_floor_f32 %3, %3\nv_floor_f32 %4, %4\nv_floor_f32 %5, %5\nv_floor_f32 %6, %6\nv_floor_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 15) { // v_cvt_i32_f32
                asm volatile("v_cvt_i32_f32 %0, %0\nv_cvt_i32_f32 %1, %1\nv_cvt_i32_f32 %2, %2\nv_cvt_i32_f32 %3, %3\nv_cvt_i32_f32 %4, %4\nv_cvt_i32_f32 %5, %5\nv_cvt_i32_f32 %6, %6\nv_cvt_i32_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 8) { // v_med3_f32
                asm volatile("v_med3_f32 %0, %0, %8, %9\nv_med3_f32 %1, %1, %8, %9\nv_med3_f32 %2, %2, %8, %9\nv_med3_f32 %3, %3, %8, %9\nv_med3_f32 %4, %4, %8, %9\nv_med3_f32 %5, %5, %8, %9\nv_med3_f32 %6, %6, %8, %9\nv_med3_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd)
{
    float* out; long long* cyc;
    const int blocks = 256, threads = 256 * waves_per_simd;
    hipMalloc(&out, blocks * threads * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    const int iters = 200;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double ninstr = (double)iters * REP;   // per wave
    // wall-clock based: instr per SIMD = ninstr * waves_per_simd
    printf("%-22s waves/SIMD %d : %.3f ms, s_memtime ticks/instr/wave %.2f, ns per SIMD-instr %.3f\n", name,
           waves_per_simd, ms, (double)h[0] / ninstr, ms * 1e6 / (ninstr * waves_per_simd));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w = 1; w <= 4; w++) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_cndmask_b32", w); run<3>("v_mov_b32_dpp", w);
        run<4>("v_rcp_f32", w); run<5>("v_add_f32", w); run<6>("cvt+lshl_add", w); run<7>("v_fma_f64", w); run<8>("v_med3_f32", w);
        run<9>("v_cndmask_e64_sgpr", w); run<10>("v_fmac_f32", w); run<11>("v_fma_f32 dependent", w);
        run<12>("v_fract_f32", w); run<13>("v_cvt_flr_i32_f32", w); run<14>("v_floor_f32", w); run<15>("v_cvt_i32_f32", w);
    }
    return 0;
}
