#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int t = threadIdx.x; t < 4096; t += 64) lds[t] = (unsigned short)t;
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    // image: [voxel][16 channels]; group g covers voxels 4g..4g+3; source lane i -> row i>>2, channel quad i&3
    const unsigned short* ap = lds + (4 * g + (i >> 2)) * 16 + (i & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)ap);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int want = (4 * (l >> 4) + j) * 16 + (l & 15); if (h[l * 4 + j] != want) { if (bad < 8) printf("lane %d j %d got %d want %d\n", l, j, h[l*4+j], want); ++bad; } }
    printf("bad = %d\n", bad);
    for (int l = 0; l < 20; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
