// Checks the row all-reduce helpers of common.h (v_permlane16_swap / v_permlane32_swap) against the shuffle form they replace.
// hipcc --offload-arch=gfx950 -O3 -I../../deepatlas_amd/csrc permlane_swap.hip -o /tmp/permlane_swap && /tmp/permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
#include "common.h"
__global__ void k(const float* in, float* o) {
    const float v = in[threadIdx.x];
    float s = v; s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    float m = v; m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    o[threadIdx.x] = da_rows_sum(v) - s;
    o[64 + threadIdx.x] = da_rows_max(v) - m;
    da_u32x2 r = __builtin_amdgcn_permlane16_swap(threadIdx.x, 100 + threadIdx.x, false, false);
    o[128 + threadIdx.x] = (float)r[0]; o[192 + threadIdx.x] = (float)r[1];
}
int main() {
    float h[64], *d, *o, ho[256];
    for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64) * 0.37f - 5.f;
    hipMalloc(&d, 256); hipMalloc(&o, 1024);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    float es = 0, em = 0; for (int i = 0; i < 64; ++i) { es = fmaxf(es, fabsf(ho[i])); em = fmaxf(em, fabsf(ho[64 + i])); }
    printf("max |sum diff| %g  max |max diff| %g\n", es, em);
    printf("permlane16_swap(lane, 100 + lane): r0 lanes 0,16,32,48 = %g %g %g %g ; r1 = %g %g %g %g\n", ho[128], ho[144], ho[160], ho[176], ho[192], ho[208], ho[224], ho[240]);
    return 0;
}
