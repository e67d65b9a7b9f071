// How expensive is v_cmp + v_cndmask really?  (e32/VCC vs e64/SGPR forms)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int OP>
__global__ void k(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const float b = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 4; r++) {
            if (OP == 0) {   // cmp -> vcc, cndmask e32 reads vcc   (what the compiler emits)
                asm volatile("v_cmp_gt_f32 vcc, %0, %4\nv_cndmask_b32 %0, %0, %5, vcc\nv_cmp_gt_f32 vcc, %1, %4\nv_cndmask_b32 %1, %1, %5, vcc\n"
                             "v_cmp_gt_f32 vcc, %2, %4\nv_cndmask_b32 %2, %2, %5, vcc\nv_cmp_gt_f32 vcc, %3, %4\nv_cndmask_b32 %3, %3, %5, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
            } else if (OP == 1) {   // cmp -> sgpr pair, cndmask e64
                asm volatile("v_cmp_gt_f32 s[10:11], %0, %4\nv_cndmask_b32_e64 %0, %0, %5, s[10:11]\nv_cmp_gt_f32 s[12:13], %1, %4\nv_cndmask_b32_e64 %1, %1, %5, s[12:13]\n"
                             "v_cmp_gt_f32 s[14:15], %2, %4\nv_cndmask_b32_e64 %2, %2, %5, s[14:15]\nv_cmp_gt_f32 s[16:17], %3, %4\nv_cndmask_b32_e64 %3, %3, %5, s[16:17]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
            } else if (OP == 2) {   // one cmp, 7 cndmask e32 on vcc
                asm volatile("v_cmp_gt_f32 vcc, %0, %4\nv_cndmask_b32 %0, %0, %5, vcc\nv_cndmask_b32 %1, %1, %5, vcc\nv_cndmask_b32 %2, %2, %5, vcc\nv_cndmask_b32 %3, %3, %5, vcc\n"
                             "v_cndmask_b32 %0, %0, %5, vcc\nv_cndmask_b32 %1, %1, %5, vcc\nv_cndmask_b32 %2, %2, %5, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
            } else if (OP == 3) {   // v_max_f32 x8 (select-free alternative)
                asm volatile("v_max_f32 %0, %0, %4\nv_max_f32 %1, %1, %4\nv_max_f32 %2, %2, %4\nv_max_f32 %3, %3, %4\nv_max_f32 %0, %0, %5\nv_max_f32 %1, %1, %5\nv_max_f32 %2, %2, %5\nv_max_f32 %3, %3, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
            } else if (OP == 4) {   // v_cmp only -> vcc x8
                asm volatile("v_cmp_gt_f32 vcc, %0, %4\nv_cmp_gt_f32 vcc, %1, %4\nv_cmp_gt_f32 vcc, %2, %4\nv_cmp_gt_f32 vcc, %3, %4\nv_cmp_gt_f32 vcc, %0, %5\nv_cmp_gt_f32 vcc, %1, %5\nv_cmp_gt_f32 vcc, %2, %5\nv_cmp_gt_f32 vcc, %3, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
            } else if (OP == 5) {   // v_and_or_b32 x8
                asm volatile("v_and_or_b32 %0, %0, %4, %5\nv_and_or_b32 %1, %1, %4, %5\nv_and_or_b32 %2, %2, %4, %5\nv_and_or_b32 %3, %3, %4, %5\nv_and_or_b32 %0, %0, %4, %5\nv_and_or_b32 %1, %1, %4, %5\nv_and_or_b32 %2, %2, %4, %5\nv_and_or_b32 %3, %3, %4, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int OP> void run(const char* name, int w)
{
    float* out; const int blocks = 256, threads = 256 * w; hipMalloc(&out, blocks * threads * 4);
    const int iters = 200;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s w=%d ns per SIMD-instr %.3f\n", name, w, ms * 1e6 / ((double)iters * REP * 2 * w));
    hipFree(out);
}
int main() { for (int w = 1; w <= 3; w += 1) { run<0>("cmp->vcc + cndmask_e32 (pairs)", w); run<1>("cmp->sgpr + cndmask_e64 (pairs)", w); run<2>("1 cmp + 7 cndmask_e32", w); run<3>("v_max_f32", w); run<4>("v_cmp_gt_f32 vcc", w); run<5>("v_and_or_b32", w);} }
