// Which workgroups of a 512-block launch (256 threads, 60 KB of LDS: two per CU) share a CU?  Prints, per block, (XCC, SE, SH, CU) from
// HW_REG_HW_ID / HW_REG_XCC_ID and the block ids resident on each CU.   hipcc --offload-arch=gfx950 -O3 -o wg_placement wg_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256, 2) probe(unsigned* out, int spin) {
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = (float)hw;
    const long long t0 = clock64();
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    if (lds[(threadIdx.x + 1) & 255] < -1.f) out[0] = 0;
}
int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 512;
    unsigned* d; hipMalloc(&d, nb * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 60 * 1024, 0, d, 200000);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(2 * nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
        const unsigned cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cuid].push_back(b);
        if (b < 24) printf("block %3d: xcc %u se %u sh %u cu %2u  (hw_id %08x)\n", b, xcc, se, sh, cuid, hw);
    }
    printf("%zu distinct (xcc, se, sh, cu)\n", cu.size());
    int shown = 0; std::map<int, int> delta;
    for (auto& kv : cu) {
        if (shown++ < 20) { printf("cu %04x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
        if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
    }
    for (auto& kv : delta) printf("pairs with id difference %d: %d\n", kv.first, kv.second);
    return 0;
}
