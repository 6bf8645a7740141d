// How many 256-thread workgroups with L bytes of dynamic LDS are really co-resident on a CU?  Every workgroup records its start time, spins
// ~200 us and records its end time; workgroups whose start lies before the first end are the resident set.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_residency.hip -o tools/ubench/lds_residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void __launch_bounds__(256, 3) k(unsigned long long* __restrict__ t, int spin) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    float a = lds[(threadIdx.x * 7) & 255];
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) a = a * 1.0001f + 0.5f;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() + (a == 1234.f); }
}
int main() {
    const int nb = 1536;
    unsigned long long* d; hipMalloc(&d, nb * 16);
    for (int ldsb : {16384, 40960, 49152, 51840, 52864, 53248, 54272, 61440, 65536, 69120, 81920}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
        int api = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k, 256, ldsb);
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), ldsb, 0, d, 20000);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * nb); hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
        unsigned long long first_end = ~0ull; for (int b = 0; b < nb; ++b) first_end = std::min(first_end, h[2 * b + 1]);
        int resident = 0; for (int b = 0; b < nb; ++b) resident += h[2 * b] < first_end;
        printf("LDS %6d B / workgroup: occupancy API %d per CU, measured %d resident of %d = %.2f per CU\n", ldsb, api, resident, nb, resident / 256.0);
    }
    return 0;
}
