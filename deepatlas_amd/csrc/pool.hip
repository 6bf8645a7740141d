// MaxPool3d(2) forward/backward and nearest-neighbour up-sampling forward/backward (NDHWC).
// Rows a2 / a8 of SURVEY.md §8: nn.MaxPool3d(2) at unets.py:230,267; F.interpolate(x, size=...) (nearest)
// at voxel_morph.py:72,74,76,80.  HBM-bound gathers: one 16-byte channel quad per lane.
#include "common.h"

namespace {

template <int VEC, typename T = float>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                    int N, int D, int H, int W, int C) {
    const int Do = D / 2, Ho = H / 2, Wo = W / 2, cq = C / VEC;
    const long long total = (long long)N * Do * Ho * Wo * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int q; long long v; da_divmod(i, cq, v, q);
        int n, od, oh, ow; da_vox4(v, Do, Ho, Wo, n, od, oh, ow);
        float m[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int d = od * 2 + (t >> 2), h = oh * 2 + ((t >> 1) & 1), w = ow * 2 + (t & 1);
            const long long po = ((((long long)n * D + d) * H + h) * W + w) * C + q * VEC;
            if (VEC == 4) {
                const float4 a = da_ldq(x, po >> 2);
                // PyTorch: update if (val > max) || isnan(val)
                m[0] = (a.x > m[0] || a.x != a.x) ? a.x : m[0]; m[1] = (a.y > m[1] || a.y != a.y) ? a.y : m[1];
                m[2] = (a.z > m[2] || a.z != a.z) ? a.z : m[2]; m[3] = (a.w > m[3] || a.w != a.w) ? a.w : m[3];
            } else {
                const float a = da_ld1(x, po); m[0] = (a > m[0] || a != a) ? a : m[0];
            }
        }
        const long long oo = ((((long long)n * Do + od) * Ho + oh) * Wo + ow) * C + q * VEC;
        if (VEC == 4) da_stq(y, oo >> 2, make_float4(m[0], m[1], m[2], m[3]));
        else da_st1(y, oo, m[0]);
    }
}

// MaxPool3d(2) whose input is a RAW producer output (deferred BatchNorm + activation, ops.LazyAct): one pass reads the raw window,
// writes the activated voxels (the skip tensor of unets.py:266, which has to exist anyway) and their maximum.  Same arithmetic as
// bn_act_fwd_kernel followed by maxpool2_fwd_kernel: act(z) = max(z, z * s) with s in [0, 1) (s = 1: no activation).  Even D, H, W.
template <typename T>
__global__ void maxpool2_fwd_pro_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, float s,
                                        T* __restrict__ act, T* __restrict__ y, int N, int D, int H, int W, int C) {
    const int Do = D / 2, Ho = H / 2, Wo = W / 2, cq = C / 4;
    const long long total = (long long)N * Do * Ho * Wo * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int q; long long v; da_divmod(i, cq, v, q);
        int n, od, oh, ow; da_vox4(v, Do, Ho, Wo, n, od, oh, ow);
        const float4 sc = reinterpret_cast<const float4*>(scale)[q], sf = reinterpret_cast<const float4*>(shift)[q];
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        float4 a[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int d = od * 2 + (t >> 2), h = oh * 2 + ((t >> 1) & 1), w = ow * 2 + (t & 1);
            a[t] = da_ldq(x, (((((long long)n * D + d) * H + h) * W + w) * C + q * 4) >> 2);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int d = od * 2 + (t >> 2), h = oh * 2 + ((t >> 1) & 1), w = ow * 2 + (t & 1);
            float4 z;
            z.x = a[t].x * sc.x + sf.x; z.y = a[t].y * sc.y + sf.y; z.z = a[t].z * sc.z + sf.z; z.w = a[t].w * sc.w + sf.w;
            z.x = fmaxf(z.x, z.x * s); z.y = fmaxf(z.y, z.y * s); z.z = fmaxf(z.z, z.z * s); z.w = fmaxf(z.w, z.w * s);
            if constexpr (DaEl<T>::bf) z = da_unpack_bf16x4(da_pack_bf16x4(z));     // the maximum is taken over the STORED (rounded) values
            da_stq(act, (((((long long)n * D + d) * H + h) * W + w) * C + q * 4) >> 2, z);
            m[0] = (z.x > m[0] || z.x != z.x) ? z.x : m[0]; m[1] = (z.y > m[1] || z.y != z.y) ? z.y : m[1];
            m[2] = (z.z > m[2] || z.z != z.z) ? z.z : m[2]; m[3] = (z.w > m[3] || z.w != z.w) ? z.w : m[3];
        }
        da_stq(y, (((((long long)n * Do + od) * Ho + oh) * Wo + ow) * C + q * 4) >> 2, make_float4(m[0], m[1], m[2], m[3]));
    }
}

// One thread per OUTPUT window and channel group: recompute the arg-max (first maximum in scan order),
// write all eight input-gradient positions (dense stores, no atomics).  Voxels of odd trailing planes
// (floor mode) are zeroed by the caller's memset when D/H/W are odd.
template <int VEC, typename T = float>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx,
                                    int N, int D, int H, int W, int C, const T* __restrict__ add) {
    const int Do = D / 2, Ho = H / 2, Wo = W / 2, cq = C / VEC;
    const long long total = (long long)N * Do * Ho * Wo * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int q; long long v; da_divmod(i, cq, v, q);
        int n, od, oh, ow; da_vox4(v, Do, Ho, Wo, n, od, oh, ow);
        float m[VEC], g[VEC]; int am[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { m[j] = -INFINITY; am[j] = 0; }
        const long long go = ((((long long)n * Do + od) * Ho + oh) * Wo + ow) * C + q * VEC;
        if (VEC == 4) { const float4 a = da_ldq(dy, go >> 2); g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; }
        else g[0] = da_ld1(dy, go);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int d = od * 2 + (t >> 2), h = oh * 2 + ((t >> 1) & 1), w = ow * 2 + (t & 1);
            const long long po = ((((long long)n * D + d) * H + h) * W + w) * C + q * VEC;
            float a[VEC];
            if (VEC == 4) { const float4 b = da_ldq(x, po >> 2); a[0] = b.x; a[1] = b.y; a[2] = b.z; a[3] = b.w; }
            else a[0] = da_ld1(x, po);
#pragma unroll
            for (int j = 0; j < VEC; ++j) if (a[j] > m[j] || a[j] != a[j]) { m[j] = a[j]; am[j] = t; }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int d = od * 2 + (t >> 2), h = oh * 2 + ((t >> 1) & 1), w = ow * 2 + (t & 1);
            const long long off = ((((long long)n * D + d) * H + h) * W + w) * C + q * VEC;
            // `add`: the gradient that reaches x through its other consumer (the skip connection), fused here instead of a
            // separate accumulation pass by autograd
            if (VEC == 4) {
                float4 o = make_float4(am[0] == t ? g[0] : 0.f, am[1] == t ? g[1] : 0.f, am[2] == t ? g[2] : 0.f, am[3] == t ? g[3] : 0.f);
                if (add) { const float4 e = da_ldq(add, off >> 2); o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
                da_stq(dx, off >> 2, o);
            } else da_st1(dx, off, ((am[0] == t) ? g[0] : 0.f) + (add ? da_ld1(add, off) : 0.f));
        }
    }
}

// The same with the pooled tensor given as the RAW producer output it was computed from (deferred BatchNorm + activation, maxpool2_fwd_pro_kernel):
// the activated values are formed again with that kernel's own expression (so the arg-max is the one the stored skip tensor has), and -- dx and the
// raw value being in registers -- the kernel also accumulates the PRODUCER's BatchNorm-backward sums, bst[workgroup][2][C] doubles =
// (sum dz, sum dz (x - mean)), dz = dx act'(x scale + shift): the reduction pass over (dx, x) that BatchNorm's backward would start with is not
// needed (da_bn_act_bwd_dbias_pre).  fp32 tensors, C / 4 a power of two <= 64 (a lane keeps its channel quad for the whole launch), even D, H, W.
__global__ void __launch_bounds__(256) maxpool2_bwd_bst_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                                               int N, int D, int H, int W, int C, const float* __restrict__ add,
                                                               const float* __restrict__ par, float slope, double* __restrict__ bst) {
    __shared__ double sred[4][2][64 * 4];
    const int Do = D / 2, Ho = H / 2, Wo = W / 2, cq = C / 4;
    const long long total = (long long)N * Do * Ho * Wo * cq;
    const int q = (int)(threadIdx.x & (unsigned)(cq - 1));           // gridDim.x * 256 is a multiple of cq: the same quad in every iteration
    const float4 mu = reinterpret_cast<const float4*>(par)[q], sc = reinterpret_cast<const float4*>(par + 2 * C)[q], sf = reinterpret_cast<const float4*>(par + 3 * C)[q];
    const float s = slope < 0.f ? 1.f : slope;                    // the forward's max(z, z s)
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    double d1[4] = {0.0, 0.0, 0.0, 0.0}, d2[4] = {0.0, 0.0, 0.0, 0.0};
    int cnt = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long v = i / cq;
        int n, od, oh, ow; da_vox4(v, Do, Ho, Wo, n, od, oh, ow);
        const float4 g4 = da_ldq(dy, (((((long long)n * Do + od) * Ho + oh) * Wo + ow) * C + q * 4) >> 2);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        float4 a[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int d = od * 2 + (t >> 2), h = oh * 2 + ((t >> 1) & 1), w = ow * 2 + (t & 1);
            a[t] = da_ldq(x, (((((long long)n * D + d) * H + h) * W + w) * C + q * 4) >> 2);
        }
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}; int am[4] = {0, 0, 0, 0};
        float4 zz[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float4 z;
            z.x = a[t].x * sc.x + sf.x; z.y = a[t].y * sc.y + sf.y; z.z = a[t].z * sc.z + sf.z; z.w = a[t].w * sc.w + sf.w;
            zz[t] = z;
            z.x = fmaxf(z.x, z.x * s); z.y = fmaxf(z.y, z.y * s); z.z = fmaxf(z.z, z.z * s); z.w = fmaxf(z.w, z.w * s);
            const float e[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) if (e[j] > m[j] || e[j] != e[j]) { m[j] = e[j]; am[j] = t; }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int d = od * 2 + (t >> 2), h = oh * 2 + ((t >> 1) & 1), w = ow * 2 + (t & 1);
            const long long off = ((((long long)n * D + d) * H + h) * W + w) * C + q * 4;
            float4 o = make_float4(am[0] == t ? g[0] : 0.f, am[1] == t ? g[1] : 0.f, am[2] == t ? g[2] : 0.f, am[3] == t ? g[3] : 0.f);
            if (add) { const float4 e = da_ldq(add, off >> 2); o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
            da_stq(dx, off >> 2, o);
            const float dz0 = o.x * da_act_grad(zz[t].x, slope), dz1 = o.y * da_act_grad(zz[t].y, slope), dz2 = o.z * da_act_grad(zz[t].z, slope), dz3 = o.w * da_act_grad(zz[t].w, slope);
            s1[0] += dz0; s1[1] += dz1; s1[2] += dz2; s1[3] += dz3;
            s2[0] += dz0 * (a[t].x - mu.x); s2[1] += dz1 * (a[t].y - mu.y); s2[2] += dz2 * (a[t].z - mu.z); s2[3] += dz3 * (a[t].w - mu.w);
        }
        if (++cnt == 8) {                                         // fp32 over 64 voxels, double beyond
#pragma unroll
            for (int j = 0; j < 4; ++j) { d1[j] += (double)s1[j]; d2[j] += (double)s2[j]; s1[j] = 0.f; s2[j] = 0.f; }
            cnt = 0;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double u = d1[j] + (double)s1[j], w2 = d2[j] + (double)s2[j];
        for (int o = cq; o < 64; o <<= 1) { u += __shfl_xor(u, o); w2 += __shfl_xor(w2, o); }      // lanes with the same quad
        if (lane < cq) { sred[wave][0][lane * 4 + j] = u; sred[wave][1][lane * 4 + j] = w2; }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * C) {
        const int k = (int)threadIdx.x / C, c = (int)threadIdx.x % C;
        bst[((size_t)blockIdx.x * 2 + k) * C + c] = (sred[0][k][c] + sred[1][k][c]) + (sred[2][k][c] + sred[3][k][c]);
    }
}

// PyTorch nearest (legacy "nearest", not "nearest-exact"): src = min(floor(dst * (float)in/out), in-1)
__device__ __forceinline__ int nearest_src(int dst, int in, int out, float scale) {
    if (in == out) return dst;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

template <int VEC, typename T = float>
__global__ void upsample_nearest_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                            int N, int D, int H, int W, int C, int Do, int Ho, int Wo) {
    const int cq = C / VEC;
    const float sd = (float)D / (float)Do, sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
    const long long total = (long long)N * Do * Ho * Wo * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int q; long long v; da_divmod(i, cq, v, q);
        int n, od, oh, ow; da_vox4(v, Do, Ho, Wo, n, od, oh, ow);
        const int d = nearest_src(od, D, Do, sd), h = nearest_src(oh, H, Ho, sh), w = nearest_src(ow, W, Wo, sw);
        const long long po = ((((long long)n * D + d) * H + h) * W + w) * C + q * VEC;
        if (VEC == 4) da_stq(y, i, da_ldq(x, po >> 2));
        else da_st1(y, i, da_ld1(x, po));
    }
}

// gather form of the backward: every input voxel sums the output voxels that map onto it.
template <int VEC, typename T = float>
__global__ void upsample_nearest_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx,
                                            int N, int D, int H, int W, int C, int Do, int Ho, int Wo) {
    const int cq = C / VEC;
    const float sd = (float)D / (float)Do, sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
    const long long total = (long long)N * D * H * W * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int q; long long v; da_divmod(i, cq, v, q);
        int n, d, h, w; da_vox4(v, D, H, W, n, d, h, w);
        // candidate output range per axis: [floor(i*out/in) - 1, floor((i+1)*out/in) + 1]
        const int d0 = max(0, (int)((long long)d * Do / D) - 1), d1 = min(Do - 1, (int)((long long)(d + 1) * Do / D) + 1);
        const int h0 = max(0, (int)((long long)h * Ho / H) - 1), h1 = min(Ho - 1, (int)((long long)(h + 1) * Ho / H) + 1);
        const int w0 = max(0, (int)((long long)w * Wo / W) - 1), w1 = min(Wo - 1, (int)((long long)(w + 1) * Wo / W) + 1);
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        for (int od = d0; od <= d1; ++od) {
            if (nearest_src(od, D, Do, sd) != d) continue;
            for (int oh = h0; oh <= h1; ++oh) {
                if (nearest_src(oh, H, Ho, sh) != h) continue;
                for (int ow = w0; ow <= w1; ++ow) {
                    if (nearest_src(ow, W, Wo, sw) != w) continue;
                    const long long po = ((((long long)n * Do + od) * Ho + oh) * Wo + ow) * C + q * VEC;
                    if (VEC == 4) { const float4 a = da_ldq(dy, po >> 2); acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; }
                    else acc[0] += da_ld1(dy, po);
                }
            }
        }
        if (VEC == 4) da_stq(dx, i, make_float4(acc[0], acc[1], acc[2], acc[3]));
        else da_st1(dx, i, acc[0]);
    }
}

// nn.Upsample(scale_factor=2, mode='trilinear') (align_corners=False; unets.py:236, the generator's upsample=True option):
// src = max((dst + 0.5) / 2 - 0.5, 0); i0 = floor(src); i1 = min(i0 + 1, in - 1); lambda = src - i0.
__device__ __forceinline__ void tri2_src(int dst, int in, int& i0, int& i1, float& l1) {
    float s = ((float)dst + 0.5f) * 0.5f - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s; if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + 1 < in ? i0 + 1 : in - 1;
    l1 = s - (float)i0;
}

template <int VEC>
__global__ void upsample_tri2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int D, int H, int W, int C) {
    const int cq = C / VEC, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)N * Do * Ho * Wo * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int q; long long v; da_divmod(i, cq, v, q);
        int n, od, oh, ow; da_vox4(v, Do, Ho, Wo, n, od, oh, ow);
        int d0, d1, h0, h1, w0, w1; float ld, lh, lw;
        tri2_src(od, D, d0, d1, ld); tri2_src(oh, H, h0, h1, lh); tri2_src(ow, W, w0, w1, lw);
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dd = (k & 4) ? d1 : d0, hh = (k & 2) ? h1 : h0, ww = (k & 1) ? w1 : w0;
            // same association as ATen's upsample_trilinear3d: w-lerp, then h, then d
            const float wgt = ((k & 4) ? ld : 1.f - ld) * (((k & 2) ? lh : 1.f - lh) * ((k & 1) ? lw : 1.f - lw));
            const float* p = x + ((((long long)n * D + dd) * H + hh) * W + ww) * C + q * VEC;
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += wgt * p[j];
        }
        float* o = y + i * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = acc[j];
    }
}

// gather form of the backward: input voxel (d, h, w) collects from the outputs whose i0 or i1 is it (<= 5 candidates per axis)
template <int VEC>
__global__ void upsample_tri2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int D, int H, int W, int C) {
    const int cq = C / VEC, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)N * D * H * W * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int q; long long v; da_divmod(i, cq, v, q);
        int n, d, h, w; da_vox4(v, D, H, W, n, d, h, w);
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        for (int od = max(0, 2 * d - 2); od <= min(Do - 1, 2 * d + 2); ++od) {
            int a0, a1; float la; tri2_src(od, D, a0, a1, la);
            const float wd = (a0 == d ? 1.f - la : 0.f) + (a1 == d ? la : 0.f);
            if (wd == 0.f) continue;
            for (int oh = max(0, 2 * h - 2); oh <= min(Ho - 1, 2 * h + 2); ++oh) {
                int b0, b1; float lb; tri2_src(oh, H, b0, b1, lb);
                const float wh = (b0 == h ? 1.f - lb : 0.f) + (b1 == h ? lb : 0.f);
                if (wh == 0.f) continue;
                for (int ow = max(0, 2 * w - 2); ow <= min(Wo - 1, 2 * w + 2); ++ow) {
                    int c0, c1; float lc; tri2_src(ow, W, c0, c1, lc);
                    const float ww = (c0 == w ? 1.f - lc : 0.f) + (c1 == w ? lc : 0.f);
                    if (ww == 0.f) continue;
                    const float wgt = wd * (wh * ww);
                    const float* p = dy + ((((long long)n * Do + od) * Ho + oh) * Wo + ow) * C + q * VEC;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] += wgt * p[j];
                }
            }
        }
        float* o = dx + i * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = acc[j];
    }
}

}  // namespace

#define DA_VEC_DISPATCH(KERNEL, total, ...)                                                                      \
    do {                                                                                                          \
        if (C % 4 == 0) hipLaunchKernelGGL((KERNEL<4>), dim3(da_grid((total) / 4, 256)), dim3(256), 0, da_stream(stream), __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<1>), dim3(da_grid((total), 256)), dim3(256), 0, da_stream(stream), __VA_ARGS__);                 \
        DA_LAUNCH_CHECK();                                                                                        \
    } while (0)
#define DA_VEC_DISPATCH_T(KERNEL, T, total, ...)                                                                  \
    do {                                                                                                          \
        if (C % 4 == 0) hipLaunchKernelGGL((KERNEL<4, T>), dim3(da_grid((total) / 4, 256)), dim3(256), 0, da_stream(stream), __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<1, T>), dim3(da_grid((total), 256)), dim3(256), 0, da_stream(stream), __VA_ARGS__);                 \
        DA_LAUNCH_CHECK();                                                                                        \
    } while (0)

template <typename T> static int maxpool2_fwd_t(const T* x, T* y, int N, int D, int H, int W, int C, void* stream) {
    if (!x || !y || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0) return DA_ERR_BADARG;
    const long long total = (long long)N * (D / 2) * (H / 2) * (W / 2) * C;
    DA_VEC_DISPATCH_T(maxpool2_fwd_kernel, T, total, x, y, N, D, H, W, C);
    return 0;
}
extern "C" int da_maxpool2_fwd(const float* x, float* y, int N, int D, int H, int W, int C, void* stream) { return maxpool2_fwd_t<float>(x, y, N, D, H, W, C, stream); }
extern "C" int da_maxpool2_fwd_bf16(const void* x, void* y, int N, int D, int H, int W, int C, void* stream) { return maxpool2_fwd_t<da_bf16>((const da_bf16*)x, (da_bf16*)y, N, D, H, W, C, stream); }

template <typename T> static int maxpool2_fwd_pro_t(const T* x, const float* pro_scale, const float* pro_shift, float pro_slope, T* act, T* y,
                                                    int N, int D, int H, int W, int C, void* stream) {
    if (!x || !pro_scale || !pro_shift || !act || !y || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0) return DA_ERR_BADARG;
    if (((D | H | W) & 1) || C % 4 != 0 || pro_slope >= 1.f) return DA_ERR_UNSUPPORTED;       // odd trailing planes / odd channel counts: materialise + da_maxpool2_fwd
    const long long total = (long long)N * (D / 2) * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL((maxpool2_fwd_pro_kernel<T>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), x, pro_scale, pro_shift,
                       pro_slope < 0.f ? 1.f : pro_slope, act, y, N, D, H, W, C);
    DA_LAUNCH_CHECK();
    return 0;
}
extern "C" int da_maxpool2_fwd_pro(const float* x, const float* pro_scale, const float* pro_shift, float pro_slope, float* act, float* y,
                                   int N, int D, int H, int W, int C, void* stream) {
    return maxpool2_fwd_pro_t<float>(x, pro_scale, pro_shift, pro_slope, act, y, N, D, H, W, C, stream);
}
extern "C" int da_maxpool2_fwd_pro_bf16(const void* x, const float* pro_scale, const float* pro_shift, float pro_slope, void* act, void* y,
                                        int N, int D, int H, int W, int C, void* stream) {
    return maxpool2_fwd_pro_t<da_bf16>((const da_bf16*)x, pro_scale, pro_shift, pro_slope, (da_bf16*)act, (da_bf16*)y, N, D, H, W, C, stream);
}

// gskip == nullptr: plain backward.  Else dx = gskip + maxpool_bwd(dy): x feeds both the pool and a skip connection (unets.py:266-267,275),
// so its gradient is the sum of the two
template <typename T> static int maxpool2_bwd_t(const T* dy, const T* x, const T* gskip, T* dx, int N, int D, int H, int W, int C, void* stream) {
    if (!dy || !x || !dx || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0) return DA_ERR_BADARG;
    if ((D | H | W) & 1) {   // floor mode: trailing planes are outside every pooling window (no gradient / only the skip gradient)
        const size_t bytes = (size_t)N * D * H * W * C * sizeof(T);
        hipError_t e = gskip ? hipMemcpyAsync(dx, gskip, bytes, hipMemcpyDeviceToDevice, da_stream(stream)) : hipMemsetAsync(dx, 0, bytes, da_stream(stream));
        if (e != hipSuccess) return (int)e;
    }
    const long long total = (long long)N * (D / 2) * (H / 2) * (W / 2) * C;
    DA_VEC_DISPATCH_T(maxpool2_bwd_kernel, T, total, dy, x, dx, N, D, H, W, C, gskip);
    return 0;
}
extern "C" int da_maxpool2_bwd(const float* dy, const float* x, float* dx, int N, int D, int H, int W, int C, void* stream) {
    return maxpool2_bwd_t<float>(dy, x, nullptr, dx, N, D, H, W, C, stream);
}
extern "C" int da_maxpool2_bwd_add(const float* dy, const float* x, const float* gskip, float* dx, int N, int D, int H, int W, int C, void* stream) {
    if (!gskip) return DA_ERR_BADARG;
    return maxpool2_bwd_t<float>(dy, x, gskip, dx, N, D, H, W, C, stream);
}
// da_maxpool2_bwd[_add] on the RAW tensor the pool's forward (da_maxpool2_fwd_pro) was given, + the BatchNorm-backward sums of that tensor's producer
// (stats4 = its statistics rows [mean | rstd | scale | shift][C], slope its activation): bst[*bst_n][2][C] doubles for da_bn_act_bwd_dbias_pre.
// DA_ERR_UNSUPPORTED (odd sizes, channel counts, bst_cap < 1024): the caller runs da_maxpool2_bwd[_add] on the activated tensor and the usual backward.
extern "C" int da_maxpool2_bwd_bst(const float* dy, const float* x_raw, const float* gskip, float* dx, int N, int D, int H, int W, int C,
                                   const float* stats4, float slope, double* bst, int bst_cap, int* bst_n, void* stream) {
    if (bst_n) *bst_n = 0;
    if (!dy || !x_raw || !dx || !stats4 || !bst || !bst_n || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0) return DA_ERR_BADARG;
    const int cq = C / 4;
    if (((D | H | W) & 1) || C % 4 != 0 || cq > 64 || (cq & (cq - 1)) != 0 || 2 * C > 256 || bst_cap < 1024 || slope >= 1.f) return DA_ERR_UNSUPPORTED;
    const long long total = (long long)N * (D / 2) * (H / 2) * (W / 2) * cq;
    int nb = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(maxpool2_bwd_bst_kernel, dim3(nb), dim3(256), 0, da_stream(stream), dy, x_raw, dx, N, D, H, W, C, gskip, stats4, slope, bst);
    DA_LAUNCH_CHECK();
    *bst_n = nb;
    return 0;
}
extern "C" int da_maxpool2_bwd_bf16(const void* dy, const void* x, void* dx, int N, int D, int H, int W, int C, void* stream) {
    return maxpool2_bwd_t<da_bf16>((const da_bf16*)dy, (const da_bf16*)x, nullptr, (da_bf16*)dx, N, D, H, W, C, stream);
}
extern "C" int da_maxpool2_bwd_add_bf16(const void* dy, const void* x, const void* gskip, void* dx, int N, int D, int H, int W, int C, void* stream) {
    if (!gskip) return DA_ERR_BADARG;
    return maxpool2_bwd_t<da_bf16>((const da_bf16*)dy, (const da_bf16*)x, (const da_bf16*)gskip, (da_bf16*)dx, N, D, H, W, C, stream);
}

template <typename T> static int upsample_nearest_fwd_t(const T* x, T* y, int N, int D, int H, int W, int C, int Do, int Ho, int Wo, void* stream) {
    if (!x || !y || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return DA_ERR_BADARG;
    const long long total = (long long)N * Do * Ho * Wo * C;
    DA_VEC_DISPATCH_T(upsample_nearest_fwd_kernel, T, total, x, y, N, D, H, W, C, Do, Ho, Wo);
    return 0;
}
template <typename T> static int upsample_nearest_bwd_t(const T* dy, T* dx, int N, int D, int H, int W, int C, int Do, int Ho, int Wo, void* stream) {
    if (!dy || !dx || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return DA_ERR_BADARG;
    const long long total = (long long)N * D * H * W * C;
    DA_VEC_DISPATCH_T(upsample_nearest_bwd_kernel, T, total, dy, dx, N, D, H, W, C, Do, Ho, Wo);
    return 0;
}
extern "C" int da_upsample_nearest_fwd(const float* x, float* y, int N, int D, int H, int W, int C, int Do, int Ho, int Wo, void* stream) {
    return upsample_nearest_fwd_t<float>(x, y, N, D, H, W, C, Do, Ho, Wo, stream);
}
extern "C" int da_upsample_nearest_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C, int Do, int Ho, int Wo, void* stream) {
    return upsample_nearest_bwd_t<float>(dy, dx, N, D, H, W, C, Do, Ho, Wo, stream);
}
extern "C" int da_upsample_nearest_fwd_bf16(const void* x, void* y, int N, int D, int H, int W, int C, int Do, int Ho, int Wo, void* stream) {
    return upsample_nearest_fwd_t<da_bf16>((const da_bf16*)x, (da_bf16*)y, N, D, H, W, C, Do, Ho, Wo, stream);
}
extern "C" int da_upsample_nearest_bwd_bf16(const void* dy, void* dx, int N, int D, int H, int W, int C, int Do, int Ho, int Wo, void* stream) {
    return upsample_nearest_bwd_t<da_bf16>((const da_bf16*)dy, (da_bf16*)dx, N, D, H, W, C, Do, Ho, Wo, stream);
}

extern "C" int da_upsample_trilinear2_fwd(const float* x, float* y, int N, int D, int H, int W, int C, void* stream) {
    if (!x || !y || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return DA_ERR_BADARG;
    const long long total = (long long)N * D * H * W * 8 * C;
    DA_VEC_DISPATCH(upsample_tri2_fwd_kernel, total, x, y, N, D, H, W, C);
    return 0;
}

extern "C" int da_upsample_trilinear2_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C, void* stream) {
    if (!dy || !dx || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return DA_ERR_BADARG;
    const long long total = (long long)N * D * H * W * C;
    DA_VEC_DISPATCH(upsample_tri2_bwd_kernel, total, dy, dx, N, D, H, W, C);
    return 0;
}
