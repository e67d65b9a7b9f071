// Which CU does block b land on?  512 blocks x 256 threads, 2 blocks/CU resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int spin)
{
    __shared__ float pad[8192];   // 32 KB like the LSTM kernels
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = __builtin_readcyclecounter();
    float acc = threadIdx.x;
    for (int i = 0; i < spin; i++) { acc = acc * 1.0001f + 0.5f; pad[(threadIdx.x + i) & 8191] = acc; }
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = hwid; out[blockIdx.x * 4 + 1] = xcc; out[blockIdx.x * 4 + 2] = (unsigned)(t0 >> 8); out[blockIdx.x*4+3] = (unsigned)acc; }
}
int main()
{
    const int nb = 512;
    unsigned* d; hipMalloc(&d, nb * 16);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4);
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < nb; b++) {
        unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 0xF;
        unsigned cu_id = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu_id;
        cu[key].push_back(b);
        if (b < 40) printf("block %3d -> xcc %u se %u sh %u cu %u (hwid %08x)\n", b, xcc, se, sh, cu_id, hw);
    }
    printf("distinct CUs used: %zu\n", cu.size());
    std::map<size_t,int> hist; int shown = 0;
    for (auto& kv : cu) { hist[kv.second.size()]++; if (shown++ < 24) { printf("cu %05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); } }
    for (auto& kv : hist) printf("%d CUs host %zu blocks\n", kv.second, kv.first);
    return 0;
}
