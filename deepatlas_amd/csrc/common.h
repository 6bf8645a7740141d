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

// XCD-contiguous grid-stride loop for kernels whose neighbouring items share cache lines (stencils, gathers).  Workgroup b runs on XCD b % 8 and every
// XCD has its own L2: with the plain loop `i = b * blockDim + t; i += gridDim * blockDim` all eight XCDs walk the same window of the tensor together and
// each of their L2s fetches every line of it -- 7 - 8 x the tensor's bytes over the fabric (measured: bending energy 410 + 730 MB per call for a 59 MB
// field, profiles/r06_step_traffic_reg.txt).  Here XCD x owns the contiguous eighth [total x / 8, total (x + 1) / 8) (rounded to `align` items) and its
// gridDim / 8 workgroups stride through it together.  Falls back to the plain loop when gridDim.x is not a multiple of 8.
//   for (DaXcdLoop L = da_xcd_loop(total); L.i < L.end; L.i += L.step) { ... L.i ... }
struct DaXcdLoop { long long i, end, step; };
__device__ __forceinline__ DaXcdLoop da_xcd_loop(long long total, long long align = 256) {
    DaXcdLoop L;
    const long long G = gridDim.x, b = blockIdx.x, T = blockDim.x;
    if ((G & 7) != 0) { L.i = b * T + threadIdx.x; L.end = total; L.step = G * T; return L; }
    const long long x = b & 7, j = b >> 3, J = G >> 3;
    const long long units = (total + align - 1) / align;
    const long long lo = units * x / 8 * align, hi = (x == 7) ? total : units * (x + 1) / 8 * align;
    L.i = lo + j * T + threadIdx.x; L.end = hi < total ? hi : total; L.step = J * T;
    return L;
}

// the same split for loops over work items (tiles, rows): `for (it = b; it < n; it += G)` becomes `for (DaXcdItems L = da_xcd_items(n, ...); L.i < L.end; L.i += L.step)`.
// sub / nsub: position of a wave inside its workgroup when the waves, not the workgroups, take the items.
struct DaXcdItems { long long i, end, step; };
__device__ __forceinline__ DaXcdItems da_xcd_items(long long n, int sub = 0, int nsub = 1) {
    DaXcdItems L;
    const long long G = gridDim.x, b = blockIdx.x;
    if ((G & 7) != 0) { L.i = b * nsub + sub; L.end = n; L.step = G * nsub; return L; }
    const long long x = b & 7, j = b >> 3, J = G >> 3;
    L.i = n * x / 8 + j * nsub + sub; L.end = n * (x + 1) / 8; L.step = J * nsub;
    return L;
}
// ... and for kernels that take ONE item per workgroup (grid = item count): the item of workgroup b, consecutive items on the same XCD (bijective)
__device__ __forceinline__ int da_xcd_item_of_block(int b, int n) {
    const int q = n / 8, r = n % 8;
    const int xcd = b % 8, loc = b / 8;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
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

// ---------------------------------------------------------------------------------------------------
// bf16 ACTIVATION STORAGE (BASELINE configs[4]: "bf16 activations / MFMA, fp32 master weights, fp32 reductions").  The `_bf16` twins of
// the C-ABI entry points take the network-internal activation / gradient tensors as bf16 in HBM (same NDHWC layout, 2 bytes per
// element); everything a kernel computes with stays fp32 (or double for the reductions) and a value is rounded to nearest-even once,
// when it is stored.  Kernels are templates over the storage type T in {float, da_bf16} and touch memory only through these helpers.
// ---------------------------------------------------------------------------------------------------
struct da_bf16 { unsigned short v; };
template <typename T> struct DaEl;
template <> struct DaEl<float>   { static constexpr int bytes = 4; static constexpr bool bf = false; };
template <> struct DaEl<da_bf16> { static constexpr int bytes = 2; static constexpr bool bf = true; };

__device__ __forceinline__ unsigned da_pack_bf16x2(float lo, float hi) {      // round-to-nearest-even, one v_cvt_pk_bf16_f32 (element 0 in the low half)
    typedef float da_f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 da_bf16x2_t __attribute__((ext_vector_type(2)));
    const da_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, da_bf16x2_t));
}
__device__ __forceinline__ float da_bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float da_bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ float da_round_bf16(float v) { return da_bf16_lo(da_pack_bf16x2(v, 0.f)); }      // the value a bf16 store + load round trip yields
__device__ __forceinline__ float4 da_unpack_bf16x4(uint2 u) { return make_float4(da_bf16_lo(u.x), da_bf16_hi(u.x), da_bf16_lo(u.y), da_bf16_hi(u.y)); }
__device__ __forceinline__ uint2 da_pack_bf16x4(float4 v) { return make_uint2(da_pack_bf16x2(v.x, v.y), da_pack_bf16x2(v.z, v.w)); }

// quad (4 consecutive channels) number q of a tensor: 16-byte (fp32) / 8-byte (bf16) access
__device__ __forceinline__ float4 da_ldq(const float* p, long long q) { return reinterpret_cast<const float4*>(p)[q]; }
__device__ __forceinline__ float4 da_ldq(const da_bf16* p, long long q) { return da_unpack_bf16x4(reinterpret_cast<const uint2*>(p)[q]); }
__device__ __forceinline__ void da_stq(float* p, long long q, float4 v) { reinterpret_cast<float4*>(p)[q] = v; }
__device__ __forceinline__ void da_stq(da_bf16* p, long long q, float4 v) { reinterpret_cast<uint2*>(p)[q] = da_pack_bf16x4(v); }
// streaming forms (non-temporal: the line is not kept for this kernel's sake) for passes that touch every byte once
typedef float da_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 da_ldq_nt(const float* p, long long q) { const da_f4v v = __builtin_nontemporal_load(reinterpret_cast<const da_f4v*>(p) + q); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void da_stq_nt(float* p, long long q, float4 v) { __builtin_nontemporal_store((da_f4v){v.x, v.y, v.z, v.w}, reinterpret_cast<da_f4v*>(p) + q); }
__device__ __forceinline__ float4 da_ldq_nt(const da_bf16* p, long long q) { return da_ldq(p, q); }
__device__ __forceinline__ void da_stq_nt(da_bf16* p, long long q, float4 v) { da_stq(p, q, v); }
// single elements
__device__ __forceinline__ float da_ld1(const float* p, long long i) { return p[i]; }
__device__ __forceinline__ float da_ld1(const da_bf16* p, long long i) { return __uint_as_float((unsigned)p[i].v << 16); }
__device__ __forceinline__ void da_st1(float* p, long long i, float v) { p[i] = v; }
__device__ __forceinline__ void da_st1(da_bf16* p, long long i, float v) { p[i].v = (unsigned short)(da_pack_bf16x2(v, 0.f) & 0xFFFFu); }

// All-reduce over the four 16-lane rows of a wave (lane = 16 g + i: same i, every g) on the VALU.  v_permlane16_swap exchanges the odd
// rows of one register with the even rows of the other, v_permlane32_swap the upper half with the lower half (gfx950): called with the
// same value twice they return (even-row copy, odd-row copy) / (lower copy, upper copy), so one swap + one add is the xor-16 / xor-32
// butterfly step -- the same pairing, hence bit-identical sums, as two ds_bpermute shuffles at ~100 cycles each on the LDS path.
typedef unsigned da_u32x2 __attribute__((ext_vector_type(2)));
// (through inline asm: hipcc 7.2 folds r[0] + r[1] of __builtin_amdgcn_permlane16_swap(v, v) into r[0] + r[0] --
// tools/ubench/permlane_swap.hip; the s_nop cover the VALU-write -> swap and swap -> VALU-read hazards the compiler would have padded)
__device__ __forceinline__ void da_swap16(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void da_swap32(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float da_rows_sum(float v) {
    float a, b;
    da_swap16(v, a, b); v = a + b;
    da_swap32(v, a, b); return a + b;
}
__device__ __forceinline__ float da_rows_max(float v) {
    float a, b;
    da_swap16(v, a, b); v = fmaxf(a, b);
    da_swap32(v, a, b); return fmaxf(a, b);
}

// Linear voxel index -> coordinates.  In 32-bit arithmetic whenever the index fits: a 64-bit integer division costs the VALU roughly ten
// times the instructions of a 32-bit one, and the gather kernels (warps, label warps) did three per lane.
__device__ __forceinline__ void da_vox4(long long v, int D, int H, int W, int& n, int& d, int& h, int& w) {
    if (v < 0x7FFFFFFFLL) {
        unsigned r = (unsigned)v;
        const unsigned q1 = r / (unsigned)W; w = (int)(r - q1 * (unsigned)W);
        const unsigned q2 = q1 / (unsigned)H; h = (int)(q1 - q2 * (unsigned)H);
        const unsigned q3 = q2 / (unsigned)D; d = (int)(q2 - q3 * (unsigned)D);
        n = (int)q3;
    } else {
        long long r = v;
        w = (int)(r % W); r /= W;
        h = (int)(r % H); r /= H;
        d = (int)(r % D); n = (int)(r / D);
    }
}
// i -> (i / m, i % m) for a small positive m (lanes per voxel, channels): shift / mask for a power of two, 32-bit when the index fits
__device__ __forceinline__ void da_divmod(long long i, int m, long long& quot, int& rem) {
    if ((m & (m - 1)) == 0) { quot = i >> (__ffs(m) - 1); rem = (int)(i & (long long)(m - 1)); }
    else if (i < 0x7FFFFFFFLL) { const unsigned r = (unsigned)i, qq = r / (unsigned)m; quot = (long long)qq; rem = (int)(r - qq * (unsigned)m); }
    else { quot = i / m; rem = (int)(i % m); }
}
__device__ __forceinline__ void da_vox3(long long v, int H, int W, int& d, int& h, int& w) {
    if (v < 0x7FFFFFFFLL) {
        const unsigned r = (unsigned)v;
        const unsigned q1 = r / (unsigned)W; w = (int)(r - q1 * (unsigned)W);
        const unsigned q2 = q1 / (unsigned)H; h = (int)(q1 - q2 * (unsigned)H);
        d = (int)q2;
    } else {
        long long r = v;
        w = (int)(r % W); r /= W;
        h = (int)(r % H); d = (int)(r / H);
    }
}

