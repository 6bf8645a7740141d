// Shared device helpers for libdeepatlas_hip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/deepatlas_hip.h"

#define DA_WAVE 64

#define DA_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

static inline hipStream_t da_stream(void* s) { return (hipStream_t)s; }
__host__ __device__ static inline long long da_cdiv(long long a, long long b) { return (a + b - 1) / b; }
static inline size_t da_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// grid size for grid-stride elementwise kernels: enough workgroups to fill 256 CUs x 8, capped
static inline int da_grid(long long work_items, int block, int cap = 256 * 16) {
    long long g = da_cdiv(work_items, block);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

__device__ __forceinline__ float da_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double da_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float da_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// block-wide sum of a double; result valid in thread 0.  `red` must hold blockDim.x/64 doubles.
__device__ __forceinline__ double da_block_sum(double v, double* red) {
    v = da_wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += red[i];
    }
    return r;
}

__device__ __forceinline__ float da_act(float z, float slope) {
    // slope < 0: identity; slope == 0: ReLU; slope > 0: LeakyReLU(slope)
    return (slope < 0.f) ? z : (z > 0.f ? z : z * slope);
}
__device__ __forceinline__ float da_act_grad(float z, float slope) {
    // derivative wrt pre-activation z; PyTorch: grad * (z > 0 ? 1 : slope)
    return (slope < 0.f) ? 1.f : (z > 0.f ? 1.f : slope);
}

// out[o] = sum_b partial[b][o] in double.  256 threads = 64 consecutive outputs x 4 slices of the partial list
// (coalesced 256-byte rows, 4x shorter serial chains, O/64 workgroups), combined through LDS.
int da_reduce_partials(const float* partial, int nparts, int O, float* out, hipStream_t st);
// losses.hip: Dice loss / coefficients from per-block partial sums (shared with the fused label-warp Dice in warp.hip)
int da_dice_finish(double* partial, int nblocks, int N, int C, int weight_type, int no_bg, float eps,
                   float* loss, float* coef, float* isc, hipStream_t st);
