// Micro-benchmark: sustained v_mfma_f32_16x16x4_f32 rate with the conv kernel's instruction mix.
// variant 0: MFMAs only (8 accumulators x 4 dependent steps, m-outer order)
// variant 1: + 8 ds_read_b128 per 32 MFMAs (fresh A fragments each step, 2-way conflicted layout like the conv tile)
// variant 2: variant 1 + one global_load_dwordx4 (B fragment) per 32 MFMAs
// variant 3: variant 1 with a conflict-free LDS layout
// variant 4: variant 1 with the A fragments of iteration it+1 read while iteration it's MFMAs issue (double buffer)
// variant 5: variant 4 split in two half-steps of 4 M-tiles (the conv kernel's scheme)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_rate.hip -o /tmp/mfma_rate ; run: /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ void __launch_bounds__(256) k(const float* __restrict__ w, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    for (int t = threadIdx.x; t < 16384; t += 256) lds[t] = (float)(t & 7) * 0.001f;
    __syncthreads();
    f32x4 acc[8];
    for (int r = 0; r < 8; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a[8];
    for (int r = 0; r < 8; ++r) a[r] = (f32x4){1.f + r, 0.5f, 0.25f, 0.125f};
    f32x4 b = (f32x4){0.1f, 0.2f, 0.3f, 0.4f};
    const int stride = (VAR == 3) ? 20 : 16;                       // floats per "voxel"
    const float* abase = lds + (wave * 180 + i) * stride + g * 4;
    const f32x4* wp = reinterpret_cast<const f32x4*>(w) + lane;
    if (VAR == 4) {
        f32x4 an[8];
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) an[r] = *reinterpret_cast<const f32x4*>(abase + ((r * 18 + ((it + 1) % 3)) * stride));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][m], b[m], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] = *reinterpret_cast<const f32x4*>(abase + ((r * 18 + ((it + 2) % 3)) * stride));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[r][m], b[m], acc[r], 0, 0, 0);
        }
    } else if (VAR == 5) {
        f32x4 a0[4], a1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a0[r] = *reinterpret_cast<const f32x4*>(abase + ((r * 18) * stride));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a1[r] = *reinterpret_cast<const f32x4*>(abase + (((4 + r) * 18 + (it % 3)) * stride));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r][m], b[m], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) a0[r] = *reinterpret_cast<const f32x4*>(abase + ((r * 18 + ((it + 1) % 3)) * stride));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[4 + r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r][m], b[m], acc[4 + r], 0, 0, 0);
        }
    } else
    for (int it = 0; it < iters; ++it) {
        if (VAR >= 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] = *reinterpret_cast<const f32x4*>(abase + ((r * 18 + (it % 3)) * stride));
        }
        if (VAR == 2) b = wp[(it & 31) * 64];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][m], b[m], acc[r], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int r = 1; r < 8; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int VAR>
void run(int blocks, const float* w, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), 65536, 0, w, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), 65536, 0, w, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 32 * 2048.0;
    printf("variant %d blocks %4d: %.3f ms  %.1f TFLOP/s\n", VAR, blocks, ms, flops / ms / 1e9);
}

int main() {
    float *w, *out; hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20); hipMalloc(&out, 4096 * 256 * 4);
    for (int blocks : {256, 512}) { run<0>(blocks, w, out); run<1>(blocks, w, out); run<2>(blocks, w, out); run<3>(blocks, w, out); run<4>(blocks, w, out); run<5>(blocks, w, out); }
    return 0;
}
