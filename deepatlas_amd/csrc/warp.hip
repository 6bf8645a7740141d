// Deformation-field trilinear warp (spatial transformer) forward/backward + identity grid.
// Rows a9-a10 of SURVEY.md §8: `deform = disp + identity` (voxel_morph.py:85-88, lib/utils.py:89-102) and
// F.grid_sample(src, deform.permute(0,2,3,4,1), 'bilinear', 'zeros', align_corners=True) (voxel_morph.py:90-91).
// NDHWC: src[N][D][H][W][C], disp[N][D][H][W][3] with channel order (x, y, z) = (W, H, D) axis.
// HBM-bound gather: 12-byte disp read + 8 corner taps of C contiguous floats (16-byte lanes when C % 4 == 0).
#include "common.h"
#include <cstdlib>

namespace {

struct Taps {
    int x0, y0, z0;
    float fx0, fx1, fy0, fy1, fz0, fz1;   // f?0 = coord - floor, f?1 = floor + 1 - coord
};

__device__ __forceinline__ float id_coord(int k, int size) {
    // lib/utils.py:97: arange(size).float() / (size - 1) * 2.0 - 1
    return (float)k / (float)(size - 1) * 2.0f - 1.0f;
}

__device__ __forceinline__ Taps make_taps(float gx, float gy, float gz, int D, int H, int W) {
    // grid_sampler_unnormalize(align_corners=True): ((coord + 1) / 2) * (size - 1)
    const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    const float iz = ((gz + 1.f) / 2.f) * (float)(D - 1);
    Taps t;
    const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
    t.x0 = (int)x0; t.y0 = (int)y0; t.z0 = (int)z0;
    t.fx0 = ix - x0; t.fx1 = (x0 + 1.f) - ix;
    t.fy0 = iy - y0; t.fy1 = (y0 + 1.f) - iy;
    t.fz0 = iz - z0; t.fz1 = (z0 + 1.f) - iz;
    return t;
}

__device__ __forceinline__ bool is_finite_coord(float a, float b, float c) {
    // NaN / huge coordinates -> every tap out of range (int conversion of NaN is undefined)
    return fabsf(a) < 1e9f && fabsf(b) < 1e9f && fabsf(c) < 1e9f;
}

// LPV lanes cooperate on one voxel (each owns VEC contiguous channels); LPV == 1 loops over all channels.
template <int VEC>
__global__ void warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ disp,
                                float* __restrict__ deform, float* __restrict__ out,
                                int N, int D, int H, int W, int C, int lpv) {
    const long long nvox = (long long)N * D * H * W;
    const long long total = nvox * lpv;
    for (DaXcdLoop L = da_xcd_loop(total, 256 * (long long)lpv); L.i < L.end; L.i += L.step) {      // (a target neighbourhood gathers from one XCD's L2)
        const long long i = L.i;
        int q; long long v; da_divmod(i, lpv, v, q);
        int n, d, h, w; da_vox4(v, D, H, W, n, d, h, w);
        const float gx = disp[v * 3 + 0] + id_coord(w, W);
        const float gy = disp[v * 3 + 1] + id_coord(h, H);
        const float gz = disp[v * 3 + 2] + id_coord(d, D);
        if (deform && q == 0) { deform[v * 3 + 0] = gx; deform[v * 3 + 1] = gy; deform[v * 3 + 2] = gz; }
        const bool fin = is_finite_coord(gx, gy, gz);
        const Taps t = make_taps(fin ? gx : -4.f, fin ? gy : -4.f, fin ? gz : -4.f, D, H, W);
        const float* sb = src + (long long)n * D * H * W * C;
        if (VEC == 4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int cz = k >> 2, cy = (k >> 1) & 1, cx = k & 1;
                const int x = t.x0 + cx, y = t.y0 + cy, z = t.z0 + cz;
                if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) {
                    const float wgt = (cx ? t.fx0 : t.fx1) * (cy ? t.fy0 : t.fy1) * (cz ? t.fz0 : t.fz1);
                    const float4 a = *reinterpret_cast<const float4*>(sb + (((long long)z * H + y) * W + x) * C + q * 4);
                    acc.x += a.x * wgt; acc.y += a.y * wgt; acc.z += a.z * wgt; acc.w += a.w * wgt;
                }
            }
            *reinterpret_cast<float4*>(out + v * C + q * 4) = acc;
        } else {
            for (int c = 0; c < C; ++c) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int cz = k >> 2, cy = (k >> 1) & 1, cx = k & 1;
                    const int x = t.x0 + cx, y = t.y0 + cy, z = t.z0 + cz;
                    if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) {
                        const float wgt = (cx ? t.fx0 : t.fx1) * (cy ? t.fy0 : t.fy1) * (cz ? t.fz0 : t.fz1);
                        acc += sb[(((long long)z * H + y) * W + x) * C + c] * wgt;
                    }
                }
                out[v * C + c] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The many-channel gather, grouped (round 6).  In the kernels above the C / 4 lanes of a voxel each derive the voxel's taps themselves (index
// division, three float divisions, floor, bounds): at C = 32 that arithmetic ran eight times per voxel and, with each corner's load and its
// use inside one bounds branch, the eight loads of a voxel went out one round trip after the other -- the 32-channel warp sat at a quarter
// of what the L1 path delivers, VALU- and latency-bound.  Here a wave first derives ONE plan per lane (64 voxels: base offset of the
// (x0, y0, z0) corner, six weights, an 8-bit in-range mask), then walks the 64 voxels in `lpv` groups of 64 / lpv: the group's lanes fetch
// their voxel's plan by cross-lane reads (ds_bpermute, no memory) and issue all eight 16-byte corner loads back to back (an out-of-range
// corner re-reads element q of the sample and is not added).  Same taps, same weights, same order of the eight additions per channel.
// ------------------------------------------------------------------------------------------------
struct GatherPlan { int base, mask; float fx0, fx1, fy0, fy1, fz0, fz1; };

__device__ __forceinline__ GatherPlan gather_plan(float gx, float gy, float gz, bool live, int D, int H, int W, int C) {
    const bool fin = live && is_finite_coord(gx, gy, gz);
    const Taps t = make_taps(fin ? gx : -4.f, fin ? gy : -4.f, fin ? gz : -4.f, D, H, W);
    GatherPlan p;
    p.fx0 = t.fx0; p.fx1 = t.fx1; p.fy0 = t.fy0; p.fy1 = t.fy1; p.fz0 = t.fz0; p.fz1 = t.fz1;
    int m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int x = t.x0 + (k & 1), y = t.y0 + ((k >> 1) & 1), z = t.z0 + (k >> 2);
        if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) m |= 1 << k;
    }
    p.mask = fin ? m : 0;
    p.base = m ? ((t.z0 * H + t.y0) * W + t.x0) * C : 0;      // (only read where a mask bit is set: the cell then touches the volume and the product fits)
    return p;
}
__device__ __forceinline__ GatherPlan gather_plan_from(const GatherPlan& p, int sl) {
    GatherPlan r;
    r.base = __shfl(p.base, sl); r.mask = __shfl(p.mask, sl);
    r.fx0 = __shfl(p.fx0, sl); r.fx1 = __shfl(p.fx1, sl); r.fy0 = __shfl(p.fy0, sl); r.fy1 = __shfl(p.fy1, sl); r.fz0 = __shfl(p.fz0, sl); r.fz1 = __shfl(p.fz1, sl);
    return r;
}
// the 8-corner gather of one voxel's channel quad q (sb: the sample's first element), as two halves so that the next group's loads can be
// issued before this group's sums wait for theirs
__device__ __forceinline__ void gather_issue(const float* __restrict__ sb, const GatherPlan& g, int q, int H, int W, int C, float4* a) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int koff = (((k >> 2) * H + ((k >> 1) & 1)) * W + (k & 1)) * C;
#if defined(DA_WARP_ABL) && (DA_WARP_ABL & 1)
        const int off = ((g.mask >> k) & 1) ? ((g.base + koff) & 0xFFF) + q * 4 : q * 4;      // timing only: every gather hits the first 16 KB
#else
        const int off = ((g.mask >> k) & 1) ? g.base + koff + q * 4 : q * 4;
#endif
        a[k] = *reinterpret_cast<const float4*>(sb + off);
    }
}
__device__ __forceinline__ float4 gather_reduce(const GatherPlan& g, const float4* a) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float wgt = ((k & 1) ? g.fx0 : g.fx1) * (((k >> 1) & 1) ? g.fy0 : g.fy1) * ((k >> 2) ? g.fz0 : g.fz1);
        if ((g.mask >> k) & 1) { acc.x += a[k].x * wgt; acc.y += a[k].y * wgt; acc.z += a[k].z * wgt; acc.w += a[k].w * wgt; }
    }
    return acc;
}
// The walk over a wave's 64 plans, two groups in flight: `use(sl, acc)` receives the gathered quad of the voxel planned by lane sl.
template <class Use>
__device__ __forceinline__ void gather_walk(const float* __restrict__ sb, const GatherPlan& p, int lpv, int vpw, int gl, int q, int H, int W, int C, Use use) {
#if defined(DA_WARP_NOPIPE)
    for (int j = 0; j < lpv; ++j) {
        float4 a[8];
        const GatherPlan g = gather_plan_from(p, j * vpw + gl);
        gather_issue(sb, g, q, H, W, C, a);
        use(j * vpw + gl, gather_reduce(g, a));
    }
    return;
#endif
    float4 aA[8], aB[8];
    GatherPlan gA = gather_plan_from(p, gl), gB;
    gather_issue(sb, gA, q, H, W, C, aA);
    int j = 0;
    for (; j + 2 < lpv; j += 2) {
        gB = gather_plan_from(p, (j + 1) * vpw + gl);
        gather_issue(sb, gB, q, H, W, C, aB);
        use(j * vpw + gl, gather_reduce(gA, aA));
        gA = gather_plan_from(p, (j + 2) * vpw + gl);
        gather_issue(sb, gA, q, H, W, C, aA);
        use((j + 1) * vpw + gl, gather_reduce(gB, aB));
    }
    if (j + 1 < lpv) {
        gB = gather_plan_from(p, (j + 1) * vpw + gl);
        gather_issue(sb, gB, q, H, W, C, aB);
        use(j * vpw + gl, gather_reduce(gA, aA));
        use((j + 1) * vpw + gl, gather_reduce(gB, aB));
    } else use(j * vpw + gl, gather_reduce(gA, aA));
}
// V * C < 2^31 (element offsets in 32 bits), 8 <= C <= 32 (C / 4 = 2, 4 or 8 lanes per voxel: the forward tile of a wave is 64 x C x 4 bytes of LDS)
static bool gather_grouped_ok(int D, int H, int W, int C, int lpv) {
    const long long V = (long long)D * H * W;
    return lpv >= 2 && lpv <= 8 && (V + (long long)H * W + W + 1) * C < 0x7FFFFFF0LL;
}

// grid (blocks, N): a workgroup walks a contiguous range of one sample's voxels, 256 per iteration (64 per wave)
__global__ void __launch_bounds__(256) warp_fwd_grouped_kernel(const float* __restrict__ src, const float* __restrict__ disp,
                                                               float* __restrict__ deform, float* __restrict__ out, int D, int H, int W, int C, int lpv) {
    const int n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int V = D * H * W, vpw = 64 / lpv, q = lane % lpv, gl = lane / lpv;
    const int per = (int)(((long long)V + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int b = (gridDim.x % 8 == 0) ? da_xcd_item_of_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const long long r0 = (long long)b * per;
    const int v0 = r0 < V ? (int)r0 : V, v1 = r0 + per < V ? (int)(r0 + per) : V;
    const float* sb = src + (long long)n * V * C;
    float* ob = out + (long long)n * V * C;
    extern __shared__ float4 gtile[];                          // [4 waves][64 voxels][lpv quads]
    float4* tile = gtile + wave * 64 * lpv;
    for (int vb = v0 + wave * 64; vb < v1; vb += 256) {
        const int v = vb + lane;
        const bool live = v < v1;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (live) {
            int d, h, w; da_vox3(v, H, W, d, h, w);
            const float* u = disp + ((long long)n * V + v) * 3;
            gx = u[0] + id_coord(w, W); gy = u[1] + id_coord(h, H); gz = u[2] + id_coord(d, D);
            if (deform) { float* o = deform + ((long long)n * V + v) * 3; o[0] = gx; o[1] = gy; o[2] = gz; }
        }
        const GatherPlan p = gather_plan(gx, gy, gz, live, D, H, W, C);
        // the wave's 64 x C results pass through its LDS tile (the layout of 64 consecutive voxels in `out`) and leave as whole 1 KB rows: the
        // gather loop itself holds no store that the next group's loads would have to wait behind (stores and loads share one counter)
        gather_walk(sb, p, lpv, vpw, gl, q, H, W, C, [&](int sl, const float4 acc) { tile[sl * lpv + q] = acc; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4* og = reinterpret_cast<float4*>(ob + (long long)vb * C);
        for (int i = 0; i < lpv; ++i) {
            const int idx = i * 64 + lane;
#if defined(DA_WARP_ABL) && (DA_WARP_ABL & 2)
            if (vb + idx / lpv < v1 && tile[idx].x == 12345.678f) og[idx] = tile[idx];      // timing only: no output stores
#else
            if (vb + idx / lpv < v1) og[idx] = tile[idx];
#endif
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int VEC>
__global__ void warp_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ src,
                                const float* __restrict__ disp, float* __restrict__ d_disp, float* __restrict__ d_src,
                                int N, int D, int H, int W, int C, int lpv) {
    const long long nvox = (long long)N * D * H * W;
    const long long total = nvox * lpv;
    // total is padded by the launcher to a multiple of lpv*...; every lane of a voxel group runs the same trip count
    for (DaXcdLoop L = da_xcd_loop(total, 256 * (long long)lpv); L.i < L.end; L.i += L.step) {      // (a target neighbourhood gathers from one XCD's L2)
        const long long i = L.i;
        int q; long long v; da_divmod(i, lpv, v, q);
        int n, d, h, w; da_vox4(v, D, H, W, n, d, h, w);
        const float gx = disp[v * 3 + 0] + id_coord(w, W);
        const float gy = disp[v * 3 + 1] + id_coord(h, H);
        const float gz = disp[v * 3 + 2] + id_coord(d, D);
        const bool fin = is_finite_coord(gx, gy, gz);
        const Taps t = make_taps(fin ? gx : -4.f, fin ? gy : -4.f, fin ? gz : -4.f, D, H, W);
        const long long sbase = (long long)n * D * H * W * C;
        float gix = 0.f, giy = 0.f, giz = 0.f;
        const int c0 = (VEC == 4) ? q * 4 : 0;
        const int c1 = (VEC == 4) ? c0 + 4 : C;
        float go[VEC == 4 ? 4 : 1];
        if (VEC == 4) { const float4 g = *reinterpret_cast<const float4*>(dout + v * C + c0); go[0] = g.x; go[1] = g.y; go[2] = g.z; go[3] = g.w; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cz = k >> 2, cy = (k >> 1) & 1, cx = k & 1;
            const int x = t.x0 + cx, y = t.y0 + cy, z = t.z0 + cz;
            if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) {
                const float wx = cx ? t.fx0 : t.fx1, wy = cy ? t.fy0 : t.fy1, wz = cz ? t.fz0 : t.fz1;
                const float wgt = wx * wy * wz;
                const long long off = sbase + (((long long)z * H + y) * W + x) * C;
                float dot = 0.f;   // sum_c src[corner][c] * gOut[c]
                if (VEC == 4) {
                    if (d_disp) {
                        const float4 a = *reinterpret_cast<const float4*>(src + off + c0);
                        dot = a.x * go[0] + a.y * go[1] + a.z * go[2] + a.w * go[3];
                    }
                    if (d_src) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) atomicAdd(d_src + off + c0 + j, wgt * go[j]);
                    }
                } else {
                    for (int c = c0; c < c1; ++c) {
                        const float g = dout[v * C + c];
                        if (d_disp) dot += src[off + c] * g;
                        if (d_src) atomicAdd(d_src + off + c, wgt * g);
                    }
                }
                gix += (cx ? dot : -dot) * wy * wz;
                giy += (cy ? dot : -dot) * wx * wz;
                giz += (cz ? dot : -dot) * wx * wy;
            }
        }
        if (d_disp) {
            if (VEC == 4) {
                for (int o = 1; o < lpv; o <<= 1) { gix += __shfl_xor(gix, o); giy += __shfl_xor(giy, o); giz += __shfl_xor(giz, o); }
            }
            if (q == 0) {
                // grad wrt normalised coords: * (size - 1) / 2 ; d deform / d disp = 1
                d_disp[v * 3 + 0] = gix * ((float)(W - 1) / 2.f);
                d_disp[v * 3 + 1] = giy * ((float)(H - 1) / 2.f);
                d_disp[v * 3 + 2] = giz * ((float)(D - 1) / 2.f);
            }
        }
    }
}

// d_src only, one lane per channel: a wave instruction adds whole 4*C-byte voxel rows (64/C voxels per instruction), so the
// atomic units see full lines instead of every fourth dword.  Trilinear weights are recomputed per lane (cheap next to the
// 8 atomics); gOut is read fully coalesced.
__global__ void warp_bwd_dsrc_lane_kernel(const float* __restrict__ dout, const float* __restrict__ disp, float* __restrict__ d_src,
                                          int N, int D, int H, int W, int C) {
    const long long total = (long long)N * D * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c; long long v; da_divmod(i, C, v, c);
        int n, d, h, w; da_vox4(v, D, H, W, n, d, h, w);
        const float gx = disp[v * 3 + 0] + id_coord(w, W);
        const float gy = disp[v * 3 + 1] + id_coord(h, H);
        const float gz = disp[v * 3 + 2] + id_coord(d, D);
        if (!is_finite_coord(gx, gy, gz)) continue;
        const Taps t = make_taps(gx, gy, gz, D, H, W);
        const float g = dout[i];
        float* base = d_src + (long long)n * D * H * W * C + c;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cz = k >> 2, cy = (k >> 1) & 1, cx = k & 1;
            const int x = t.x0 + cx, y = t.y0 + cy, z = t.z0 + cz;
            if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) {
                const float wgt = (cx ? t.fx0 : t.fx1) * (cy ? t.fy0 : t.fy1) * (cz ? t.fz0 : t.fz1);
                atomicAdd(base + (((long long)z * H + y) * W + x) * C, wgt * g);
            }
        }
    }
}

// ---- deterministic d_src (parity runs).  Float atomics add in arrival order, so two runs of the same step differ in the last bits
// (and, through Adam, drift apart).  Integer addition is associative: every contribution w * g is converted to a 64-bit fixed-point
// number with a power-of-two scale derived from max|gOut| (itself order-independent) and accumulated with 64-bit integer atomics
// -- the sum is bit-identical whatever the order -- then converted back.  Resolution 2^-38 of max|gOut| per contribution (fp32 keeps
// 2^-24 of each term), headroom for 2^24 contributions per element.  Costs an 8-byte scratch element + two extra passes: a switch,
// not the default.
__global__ void absmax_kernel(const float* __restrict__ x, long long n, unsigned int* __restrict__ out_bits) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float a = fabsf(x[i]);
        m = (a == a && a <= 3.0e38f) ? fmaxf(m, a) : m;              // ignore NaN / inf
    }
    m = da_wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));      // non-negative floats order like their bit patterns
}

__device__ __forceinline__ float fixed_scale_from_max(unsigned int max_bits) {
    const float m = __uint_as_float(max_bits);
    if (!(m > 0.f)) return 1.f;
    int e; frexpf(m, &e);                                             // m = f * 2^e, f in [0.5, 1)  ->  m < 2^e
    return ldexpf(1.f, 38 - e);                                       // |w * g| * scale < 2^38
}

__global__ void warp_bwd_dsrc_fixed_kernel(const float* __restrict__ dout, const float* __restrict__ disp, unsigned long long* __restrict__ acc,
                                           const unsigned int* __restrict__ max_bits, int N, int D, int H, int W, int C) {
    const float scale = fixed_scale_from_max(max_bits[0]);
    const long long total = (long long)N * D * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c; long long v; da_divmod(i, C, v, c);
        int n, d, h, w; da_vox4(v, D, H, W, n, d, h, w);
        const float gx = disp[v * 3 + 0] + id_coord(w, W);
        const float gy = disp[v * 3 + 1] + id_coord(h, H);
        const float gz = disp[v * 3 + 2] + id_coord(d, D);
        if (!is_finite_coord(gx, gy, gz)) continue;
        const Taps t = make_taps(gx, gy, gz, D, H, W);
        const float g = dout[i];
        unsigned long long* base = acc + (long long)n * D * H * W * C + c;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cz = k >> 2, cy = (k >> 1) & 1, cx = k & 1;
            const int x = t.x0 + cx, y = t.y0 + cy, z = t.z0 + cz;
            if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) {
                const float wgt = (cx ? t.fx0 : t.fx1) * (cy ? t.fy0 : t.fy1) * (cz ? t.fz0 : t.fz1);
                const long long q = __double2ll_rn((double)(wgt * g) * (double)scale);
                atomicAdd(base + (((long long)z * H + y) * W + x) * C, (unsigned long long)q);      // two's complement wrap-around add
            }
        }
    }
}

__global__ void fixed_to_float_kernel(const unsigned long long* __restrict__ acc, const unsigned int* __restrict__ max_bits,
                                      float* __restrict__ out, long long n) {
    const double inv = 1.0 / (double)fixed_scale_from_max(max_bits[0]);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = (float)((double)(long long)acc[i] * inv);
}

// Warp of a LABEL map as if it were its one-hot encoding (the joint step's registration phase warps one-hot(seg_m) with the
// predicted field): out[v][c] = sum_k w_k [label[corner_k] == c].  The 32-channel one-hot tensor (629 MB per
// volume) is never materialised: 8 label bytes are read per voxel instead of 8 x 128 bytes.  One lane per channel.
__device__ __forceinline__ int warp_label_at(const void* lab, int label_bytes, long long i) {
    return label_bytes == 1 ? (int)((const unsigned char*)lab)[i] : (int)((const long long*)lab)[i];
}

template <int VEC>
__global__ void warp_labels_fwd_kernel(const void* __restrict__ labels, int label_bytes, const float* __restrict__ disp,
                                       float* __restrict__ out, int N, int D, int H, int W, int C) {
    const int cq = C / VEC;                                    // lanes per voxel, VEC channels each (16-byte stores for VEC = 4)
    const long long total = (long long)N * D * H * W * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c0; long long v; da_divmod(i, cq, v, c0); c0 *= VEC;
        int n, d, h, w; da_vox4(v, D, H, W, n, d, h, w);
        const float gx = disp[v * 3 + 0] + id_coord(w, W);
        const float gy = disp[v * 3 + 1] + id_coord(h, H);
        const float gz = disp[v * 3 + 2] + id_coord(d, D);
        const bool fin = is_finite_coord(gx, gy, gz);
        const Taps t = make_taps(fin ? gx : -4.f, fin ? gy : -4.f, fin ? gz : -4.f, D, H, W);
        const long long sbase = (long long)n * D * H * W;
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cz = k >> 2, cy = (k >> 1) & 1, cx = k & 1;
            const int x = t.x0 + cx, y = t.y0 + cy, z = t.z0 + cz;
            if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) {
                const float wgt = (cx ? t.fx0 : t.fx1) * (cy ? t.fy0 : t.fy1) * (cz ? t.fz0 : t.fz1);
                const int rel = warp_label_at(labels, label_bytes, sbase + ((long long)z * H + y) * W + x) - c0;
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += (rel == j) ? wgt : 0.f;
            }
        }
        float* o = out + v * C + c0;
        if (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else o[0] = acc[0];
    }
}

// d loss / d disp for the label warp: the same grid gradient as warp_bwd_kernel with sum_c src[corner][c] gOut[c] = gOut[label[corner]]
__global__ void warp_labels_bwd_kernel(const float* __restrict__ dout, const void* __restrict__ labels, int label_bytes,
                                       const float* __restrict__ disp, float* __restrict__ d_disp, int N, int D, int H, int W, int C) {
    const long long nvox = (long long)N * D * H * W;
    for (DaXcdLoop XL = da_xcd_loop(nvox); XL.i < XL.end; XL.i += XL.step) {
        const long long v = XL.i;
        int n, d, h, w; da_vox4(v, D, H, W, n, d, h, w);
        const float gx = disp[v * 3 + 0] + id_coord(w, W);
        const float gy = disp[v * 3 + 1] + id_coord(h, H);
        const float gz = disp[v * 3 + 2] + id_coord(d, D);
        const bool fin = is_finite_coord(gx, gy, gz);
        const Taps t = make_taps(fin ? gx : -4.f, fin ? gy : -4.f, fin ? gz : -4.f, D, H, W);
        const long long sbase = (long long)n * D * H * W;
        float gix = 0.f, giy = 0.f, giz = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cz = k >> 2, cy = (k >> 1) & 1, cx = k & 1;
            const int x = t.x0 + cx, y = t.y0 + cy, z = t.z0 + cz;
            if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) {
                const float wx = cx ? t.fx0 : t.fx1, wy = cy ? t.fy0 : t.fy1, wz = cz ? t.fz0 : t.fz1;
                const int lab = warp_label_at(labels, label_bytes, sbase + ((long long)z * H + y) * W + x);
                const float dot = (lab >= 0 && lab < C) ? dout[v * C + lab] : 0.f;
                gix += (cx ? dot : -dot) * wy * wz;
                giy += (cy ? dot : -dot) * wx * wz;
                giz += (cz ? dot : -dot) * wx * wy;
            }
        }
        d_disp[v * 3 + 0] = gix * ((float)(W - 1) / 2.f);
        d_disp[v * 3 + 1] = giy * ((float)(H - 1) / 2.f);
        d_disp[v * 3 + 2] = giz * ((float)(D - 1) / 2.f);
    }
}

__global__ void identity_grid_kernel(float* __restrict__ out, int D, int H, int W, int normalize) {
    const long long V = (long long)D * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long long)gridDim.x * blockDim.x) {
        int d, h, w; da_vox3(i, H, W, d, h, w);
        out[i] = normalize ? id_coord(w, W) : (float)w;           // channel 0: W axis (x)
        out[V + i] = normalize ? id_coord(h, H) : (float)h;       // channel 1: H axis (y)
        out[2 * V + i] = normalize ? id_coord(d, D) : (float)d;   // channel 2: D axis (z)
    }
}


// ------------------------------------------------------------------------------------------------
// Fused anatomy losses of the joint step: Dice of a WARPED segmentation against a label map.
// Dice's gradient with respect to its input is g[v][c] = a[c] * [St[v] == c] + b[c] (a, b = `coef` of da_dice_fwd): rank-structured.
//  * registration phase, Dice(warp(onehot(Sm), phi), onehot(St)): the warped 32-channel tensor (629 MB) is never built.  The three
//    per-class sums Dice needs follow from the 8 (weight, label) pairs of every voxel, and the gradient reaching the displacement
//    only needs g at the 8 corner labels -- computed from a, b on the fly.
//  * segmentation phase, Dice(warp(softmax(S(Im)), phi), onehot(St)): the adjoint warp (a scatter of 8 x 32 float atomics per voxel)
//    collapses to  W^T g = b[c] * A[u] + a[c] * B[u][c]  with  A = W^T 1  (one channel) and  B = W^T onehot(St)  (8 atomics per
//    voxel land in channel St[v] only): 16 atomics per voxel instead of 256.
// ------------------------------------------------------------------------------------------------
// Dice(warp(src, identity + disp), onehot(lab_t)) without writing the warped tensor: warp_fwd_kernel<4>'s gather (C / 4 lanes per voxel,
// 4 channels each) with the three Dice sums of the lane's channels accumulated in registers instead of the 16-byte store; per-block partials
// [N][gridDim.x][3][C] in the layout of dice_partial_vec_kernel (losses.hip), finished by da_dice_finish.
__global__ void __launch_bounds__(256) warp_dice_partial_kernel(const float* __restrict__ src, const float* __restrict__ disp,
                                                                const void* __restrict__ lab_t, int bt,
                                                                int D, int H, int W, int C, int lpv, double* __restrict__ partial) {
    extern __shared__ float shf[];   // [3][slots][C]
    const int n = blockIdx.y;
    const int slots = 256 / lpv;
    const int q = threadIdx.x % lpv, s = threadIdx.x / lpv;
    const long long V = (long long)D * H * W;
    const long long vpb = da_cdiv(V, (long long)gridDim.x);
    const long long v0 = (long long)((gridDim.x % 8 == 0) ? da_xcd_item_of_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * vpb;      // neighbouring ranges gather from ONE XCD's L2
    long long v1 = v0 + vpb; if (v1 > V) v1 = V;
    const float* sb = src + (long long)n * V * C;
    float aI[4] = {0, 0, 0, 0}, aS[4] = {0, 0, 0, 0}, aT[4] = {0, 0, 0, 0};
    for (long long v = v0 + s; v < v1; v += slots) {
        int d, h, w; da_vox3(v, H, W, d, h, w);
        const float* u = disp + ((long long)n * V + v) * 3;
        const float gx = u[0] + id_coord(w, W), gy = u[1] + id_coord(h, H), gz = u[2] + id_coord(d, D);
        const bool fin = is_finite_coord(gx, gy, gz);
        const Taps t = make_taps(fin ? gx : -4.f, fin ? gy : -4.f, fin ? gz : -4.f, D, H, W);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cz = k >> 2, cy = (k >> 1) & 1, cx = k & 1;
            const int x = t.x0 + cx, y = t.y0 + cy, z = t.z0 + cz;
            if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) {
                const float wgt = (cx ? t.fx0 : t.fx1) * (cy ? t.fy0 : t.fy1) * (cz ? t.fz0 : t.fz1);
                const float4 a = *reinterpret_cast<const float4*>(sb + (((long long)z * H + y) * W + x) * C + q * 4);
                acc.x += a.x * wgt; acc.y += a.y * wgt; acc.z += a.z * wgt; acc.w += a.w * wgt;
            }
        }
        const int rel = warp_label_at(lab_t, bt, (long long)n * V + v) - q * 4;
        const float pv[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float tt = (rel == j) ? 1.f : 0.f; aI[j] += pv[j] * tt; aS[j] += pv[j]; aT[j] += tt; }
    }
    float* sI = shf; float* sS = shf + (size_t)slots * C; float* sT = shf + (size_t)2 * slots * C;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sI[s * C + q * 4 + j] = aI[j]; sS[s * C + q * 4 + j] = aS[j]; sT[s * C + q * 4 + j] = aT[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double tI = 0, tS = 0, tT = 0;
        for (int k = 0; k < slots; ++k) { tI += sI[k * C + c]; tS += sS[k * C + c]; tT += sT[k * C + c]; }
        double* o = partial + (((size_t)n * gridDim.x + blockIdx.x) * 3) * C;
        o[c] = tI; o[C + c] = tS; o[2 * C + c] = tT;
    }
}

// The same sums with the grouped gather (warp_fwd_grouped_kernel): one plan per voxel, eight loads in flight per lane.
__global__ void __launch_bounds__(256) warp_dice_grouped_kernel(const float* __restrict__ src, const float* __restrict__ disp,
                                                                const void* __restrict__ lab_t, int bt,
                                                                int D, int H, int W, int C, int lpv, double* __restrict__ partial) {
    extern __shared__ float shf[];   // [3][slots][C]
    const int n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slots = 256 / lpv;
    const int V = D * H * W, vpw = 64 / lpv, q = lane % lpv, gl = lane / lpv, s = threadIdx.x / lpv;
    const int per = (int)(((long long)V + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int b = (gridDim.x % 8 == 0) ? da_xcd_item_of_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const long long r0 = (long long)b * per;
    const int v0 = r0 < V ? (int)r0 : V, v1 = r0 + per < V ? (int)(r0 + per) : V;
    const float* sb = src + (long long)n * V * C;
    float aI[4] = {0, 0, 0, 0}, aS[4] = {0, 0, 0, 0}, aT[4] = {0, 0, 0, 0};
    for (int vb = v0 + wave * 64; vb < v1; vb += 256) {
        const int v = vb + lane;
        const bool live = v < v1;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        int lab = -1;
        if (live) {
            int d, h, w; da_vox3(v, H, W, d, h, w);
            const float* u = disp + ((long long)n * V + v) * 3;
            gx = u[0] + id_coord(w, W); gy = u[1] + id_coord(h, H); gz = u[2] + id_coord(d, D);
            lab = warp_label_at(lab_t, bt, (long long)n * V + v);
        }
        const GatherPlan p = gather_plan(gx, gy, gz, live, D, H, W, C);
        gather_walk(sb, p, lpv, vpw, gl, q, H, W, C, [&](int sl, const float4 acc) {
            const int rel = __shfl(lab, sl) - q * 4;
            if (vb + sl < v1) {
                const float pv[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { const float tt = (rel == jj) ? 1.f : 0.f; aI[jj] += pv[jj] * tt; aS[jj] += pv[jj]; aT[jj] += tt; }
            }
        });
    }
    float* sI = shf; float* sS = shf + (size_t)slots * C; float* sT = shf + (size_t)2 * slots * C;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sI[s * C + q * 4 + j] = aI[j]; sS[s * C + q * 4 + j] = aS[j]; sT[s * C + q * 4 + j] = aT[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double tI = 0, tS = 0, tT = 0;
        for (int k = 0; k < slots; ++k) { tI += sI[k * C + c]; tS += sS[k * C + c]; tT += sT[k * C + c]; }
        double* o = partial + (((size_t)n * gridDim.x + blockIdx.x) * 3) * C;
        o[c] = tI; o[C + c] = tS; o[2 * C + c] = tT;
    }
}

// One lane per voxel.  The three per-class sums are histograms keyed by a label: S by the 8 corner labels of the moving map (value = the
// corner's weight), I and T by the target label (values: the weight that landed on corners carrying that label, and 1).  Label maps are
// piecewise constant, so the 64 voxels of a wave see one to three distinct keys: the wave loops over the DISTINCT keys present, sums the
// lanes that carry the key with a fixed-order butterfly and lane 0 adds the sum to the wave's double-precision table in LDS -- no
// atomics, a fixed summation order, and ~40 cross-lane operations per 64 voxels instead of 8 corners x C compares per voxel.  (The earlier
// form gave every voxel C / 8 lanes that each re-derived the taps and compared every corner label against their 8 classes: VALU-bound at
// 0.02 of the HBM rate.)
__device__ __forceinline__ float lwd_wave_sum(float v) {
    // rotations inside each row of 16 lanes (DPP: a few cycles each, where a ds_bpermute shuffle is a ~100-cycle LDS round trip and six of
    // them in a dependent chain cost more than the rest of the iteration), then the four row sums through scalar registers
#define DA_ROR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (n), 0xF, 0xF, false))
    v += DA_ROR(v, 8); v += DA_ROR(v, 4); v += DA_ROR(v, 2); v += DA_ROR(v, 1);
#undef DA_ROR
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

__global__ void __launch_bounds__(256) label_warp_dice_partial_kernel(const void* __restrict__ lab_m, int bm, const void* __restrict__ lab_t, int bt,
                                                                      const float* __restrict__ disp, int D, int H, int W, int C,
                                                                      double* __restrict__ partial /* [N][gridDim.x][3][C] : I, S, T */) {
    __shared__ double acc[4][3][64];
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 3 * 64; i += 256) (&acc[0][0][0])[i] = 0.0;
    __syncthreads();
    double* const aI = acc[wave][0];
    double* const aS = acc[wave][1];
    double* const aT = acc[wave][2];
    const long long V = (long long)D * H * W;
    const long long sb = (long long)n * V;
    const long long stride = (long long)gridDim.x * 256;
    // add `val` of the lanes whose key is the same class into table[key], one distinct key at a time (key < 0: lane takes no part)
    auto hist_add = [&](double* table, int key, float val) {
        unsigned long long todo = __ballot(key >= 0);
        while (todo) {
            const int c = __builtin_amdgcn_readlane(key, __ffsll((long long)todo) - 1);
            const bool m = key == c;
            const float s = lwd_wave_sum(m ? val : 0.f);
            if (lane == 0) table[c] += (double)s;
            todo &= ~__ballot(m);
        }
    };
    const DaXcdLoop XL = da_xcd_loop(V, 256);              // (an XCD's workgroups walk one contiguous eighth of the volume)
    for (long long base = XL.i - lane; base < XL.end; base += XL.step) {      // wave-uniform trip count
        const long long v = base + lane;
        const bool live = v < V;
        const long long vv = live ? v : V - 1;
        int d, h, w; da_vox3(vv, H, W, d, h, w);
        const float* u = disp + (sb + vv) * 3;
        const float gx = u[0] + id_coord(w, W), gy = u[1] + id_coord(h, H), gz = u[2] + id_coord(d, D);
        const bool fin = is_finite_coord(gx, gy, gz);
        const Taps t = make_taps(fin ? gx : -4.f, fin ? gy : -4.f, fin ? gz : -4.f, D, H, W);
        int tl = warp_label_at(lab_t, bt, sb + vv);
        if (!live || tl < 0 || tl >= C) tl = -1;
        int lab[8]; float wk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cz = k >> 2, cy = (k >> 1) & 1, cx = k & 1;
            const int x = t.x0 + cx, y = t.y0 + cy, z = t.z0 + cz;
            lab[k] = -1; wk[k] = 0.f;
            if (live && x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) {
                wk[k] = (cx ? t.fx0 : t.fx1) * (cy ? t.fy0 : t.fy1) * (cz ? t.fz0 : t.fz1);
                const int l = warp_label_at(lab_m, bm, sb + ((long long)z * H + y) * W + x);
                lab[k] = (l >= 0 && l < C) ? l : -1;
            }
        }
        // T and I: keyed by the target label
        float wi = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) wi += (lab[k] == tl) ? wk[k] : 0.f;          // (tl = -1 never matches a weight that counts: hist_add skips the lane)
        {
            unsigned long long todo = __ballot(tl >= 0);
            while (todo) {
                const int c = __builtin_amdgcn_readlane(tl, __ffsll((long long)todo) - 1);
                const bool m = tl == c;
                const unsigned long long mm = __ballot(m);
                const float s = lwd_wave_sum(m ? wi : 0.f);
                if (lane == 0) { aI[c] += (double)s; aT[c] += (double)__popcll(mm); }
                todo &= ~mm;
            }
        }
        // S: the corners that share the first corner's label go in one pass; the others (label boundaries) corner by corner
        const int key0 = lab[0];
        float s0 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s0 += (lab[k] == key0) ? wk[k] : 0.f;
        hist_add(aS, key0, s0);
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const int key = (lab[k] != key0) ? lab[k] : -1;
            if (__ballot(key >= 0)) hist_add(aS, key, wk[k]);
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 3 * C; idx += 256) {
        const int k = idx / C, c = idx % C;
        partial[(((size_t)n * gridDim.x + blockIdx.x) * 3 + k) * C + c] = acc[0][k][c] + acc[1][k][c] + acc[2][k][c] + acc[3][k][c];
    }
}

// d loss / d disp of Dice(warp(onehot(lab_m))): the grid gradient of warp_labels_bwd_kernel with g[corner label] formed from coef
__global__ void label_warp_dice_bwd_kernel(const void* __restrict__ lab_m, int bm, const void* __restrict__ lab_t, int bt,
                                           const float* __restrict__ disp, const float* __restrict__ coef, const float* __restrict__ dloss,
                                           float* __restrict__ d_disp, int N, int D, int H, int W, int C) {
    const long long V = (long long)D * H * W, nvox = V * N;
    const float gl = dloss[0];
    const int NC = N * C;
    for (DaXcdLoop XL = da_xcd_loop(nvox); XL.i < XL.end; XL.i += XL.step) {
        const long long v = XL.i;
        int n, d, h, w; da_vox4(v, D, H, W, n, d, h, w);
        const float gx = disp[v * 3 + 0] + id_coord(w, W);
        const float gy = disp[v * 3 + 1] + id_coord(h, H);
        const float gz = disp[v * 3 + 2] + id_coord(d, D);
        const bool fin = is_finite_coord(gx, gy, gz);
        const Taps t = make_taps(fin ? gx : -4.f, fin ? gy : -4.f, fin ? gz : -4.f, D, H, W);
        const long long sbase = (long long)n * V;
        const int tl = warp_label_at(lab_t, bt, v);
        float gix = 0.f, giy = 0.f, giz = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cz = k >> 2, cy = (k >> 1) & 1, cx = k & 1;
            const int x = t.x0 + cx, y = t.y0 + cy, z = t.z0 + cz;
            if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) {
                const float wx = cx ? t.fx0 : t.fx1, wy = cy ? t.fy0 : t.fy1, wz = cz ? t.fz0 : t.fz1;
                const int lab = warp_label_at(lab_m, bm, sbase + ((long long)z * H + y) * W + x);
                float dot = 0.f;
                if (lab >= 0 && lab < C) dot = gl * (coef[n * C + lab] * (lab == tl ? 1.f : 0.f) + coef[NC + n * C + lab]);
                gix += (cx ? dot : -dot) * wy * wz;
                giy += (cy ? dot : -dot) * wx * wz;
                giz += (cz ? dot : -dot) * wx * wy;
            }
        }
        d_disp[v * 3 + 0] = gix * ((float)(W - 1) / 2.f);
        d_disp[v * 3 + 1] = giy * ((float)(H - 1) / 2.f);
        d_disp[v * 3 + 2] = giz * ((float)(D - 1) / 2.f);
    }
}

// B[n][St[v]][u] += w over the 8 taps of every voxel v  (B zero-filled by the launcher).  B is CLASS-MAJOR ([N][C][V], one plane per class):
// label maps are piecewise constant and a registration field is smooth, so the 64 lanes of a wave (64 voxels along x) add into 64
// consecutive floats of one plane -- the atomics of one instruction fall into two or three 128-byte lines instead of 64 (voxel-major
// rows: one line per lane; 1.94 -> 0.54 ms at 160x192x160, profiles/r03_gather_kernels.txt).  A = W^T 1 needs no scatter of its own:
// every weight lands in exactly one class plane, so A[u] = sum_c B[c][u] (formed by the consumer from the values it reads anyway); voxels
// whose target label is outside [0, C) put their weights into the separate array A_extra (NULL when the caller knows there are none).
// Wave-level pre-aggregation along x: lane L (voxel x) and lane L + 1 (voxel x + 1) of a smooth field usually sample cells that are one voxel
// apart, so lane L's four cx = 1 corners ARE lane L + 1's four cx = 0 corners.  Where that holds (same label, same (y0, z0), x0 one apart, both
// finite: checked per lane pair) lane L hands its cx = 1 weights to lane L + 1 through a DPP-free shuffle and issues no atomics for them:
// four to six instead of eight atomics per voxel on a smooth field (0.54 -> 0.49 ms at 160x192x160; a random field has nothing to merge: 1.68 ms).
__global__ void warp_adjoint_labels_kernel(const void* __restrict__ lab_t, int bt, const float* __restrict__ disp,
                                           float* __restrict__ A_extra, float* __restrict__ B, int N, int D, int H, int W, int C) {
    const long long V = (long long)D * H * W, nvox = V * N;
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const DaXcdLoop XL = da_xcd_loop(nvox, 256);
    for (long long v0 = XL.i - lane; v0 < XL.end; v0 += XL.step) {      // whole waves iterate together (shuffles inside)
        const long long v = v0 + lane;
        const bool inr = v < nvox;
        int n = 0, d = 0, h = 0, w = 0;
        if (inr) da_vox4(v, D, H, W, n, d, h, w);
        const float gx = inr ? disp[v * 3 + 0] + id_coord(w, W) : 0.f;
        const float gy = inr ? disp[v * 3 + 1] + id_coord(h, H) : 0.f;
        const float gz = inr ? disp[v * 3 + 2] + id_coord(d, D) : 0.f;
        const bool live = inr && is_finite_coord(gx, gy, gz);
        const Taps t = make_taps(live ? gx : 0.f, live ? gy : 0.f, live ? gz : 0.f, D, H, W);
        const int lab = inr ? warp_label_at(lab_t, bt, v) : -1;
        const bool lok = lab >= 0 && lab < C;
        // this lane's cell, as one comparable key per (n, label, z0, y0) and the x0 beside it
        const bool inbox = live && t.z0 >= -1 && t.z0 < D && t.y0 >= -1 && t.y0 < H;      // (cells with no corner row inside the volume never merge)
        const long long key = inbox ? ((((long long)n * (C + 2) + (lok ? lab : C)) * (D + 2) + (t.z0 + 1)) * (H + 2) + (t.y0 + 1)) : -1 - (long long)lane;
        const long long pkey = __shfl_up(key, 1);
        const int px0 = __shfl_up(t.x0, 1);
        const bool take = inbox && lane > 0 && pkey == key && px0 + 1 == t.x0;      // lane - 1's cx = 1 corners are this lane's cx = 0 corners
        const bool given = __shfl_down((int)take, 1) != 0 && lane < 63;            // ... and lane + 1 took ours
        const long long sbase = (long long)n * V;
        float* plane = lok ? B + ((long long)n * C + lab) * V : (A_extra ? A_extra + sbase : nullptr);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            const int cz = k2 >> 1, cy = k2 & 1;
            const float wyz = (cy ? t.fy0 : t.fy1) * (cz ? t.fz0 : t.fz1);
            const float w0 = t.fx1 * wyz, w1 = t.fx0 * wyz;                       // cx = 0 | 1
            const float got = __shfl_up(w1, 1);
            const int y = t.y0 + cy, z = t.z0 + cz;
            const bool yz = live && plane && y >= 0 && y < H && z >= 0 && z < D;
            const long long row = ((long long)z * H + y) * W;
            const float a0 = w0 + (take ? got : 0.f);
            if (yz && t.x0 >= 0 && t.x0 < W && a0 != 0.f) atomicAdd(plane + row + t.x0, a0);
            if (yz && !given && t.x0 + 1 >= 0 && t.x0 + 1 < W && w1 != 0.f) atomicAdd(plane + row + t.x0 + 1, w1);
        }
    }
}

// The same scatter through an LDS box (round 5).  A workgroup takes a BX x BY x BZ box of TARGET voxels; a registration field moves a voxel by a
// voxel or two, so nearly all of its 8 * BX * BY * BZ weights land in the box grown by M cells on every side -- they are added there with LDS
// atomics, one copy of the grown box per label present (piecewise-constant label maps: up to K labels per box get a copy, found / claimed through a
// K-entry table), and the box is then flushed with ONE global atomic per non-zero cell, rows of consecutive floats.  A weight that falls outside the
// grown box, or whose label found no copy, goes to global memory directly as before: any field is handled, a wild one just gains nothing.
// 8 scattered global atomics per voxel become ~2 coalesced ones (boxes overlap in their margins, so the flush still has to add).
template <int BX, int BY, int BZ, int M, int K>
__global__ void __launch_bounds__(256) warp_adjoint_labels_box_kernel(const void* __restrict__ lab_t, int bt, const float* __restrict__ disp,
                                                                      float* __restrict__ A_extra, float* __restrict__ B, int N, int D, int H, int W, int C,
                                                                      int nbx, int nby, int nbz) {
    constexpr int LX = BX + 2 * M + 1, LY = BY + 2 * M + 1, LZ = BZ + 2 * M + 1, CELLS = LX * LY * LZ, NV = BX * BY * BZ;
    extern __shared__ float box[];                            // [K][CELLS]
    __shared__ int slot_label[K];
    const long long V = (long long)D * H * W;
    int b = da_xcd_item_of_block((int)blockIdx.x, (int)gridDim.x);      // neighbouring boxes on one XCD
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby; b /= nby;
    const int bz = b % nbz; const int n = b / nbz;
    const int ox = bx * BX - M, oy = by * BY - M, oz = bz * BZ - M;      // volume coordinates of box cell (0, 0, 0)
    for (int i = threadIdx.x; i < K * CELLS; i += 256) box[i] = 0.f;
    if (threadIdx.x < K) slot_label[threadIdx.x] = -1;
    __syncthreads();
    const long long sbase = (long long)n * V;
#pragma unroll 1
    for (int l = threadIdx.x; l < NV; l += 256) {
        const int x = bx * BX + l % BX, y = by * BY + (l / BX) % BY, z = bz * BZ + l / (BX * BY);
        if (x >= W || y >= H || z >= D) continue;
        const long long v = sbase + ((long long)z * H + y) * W + x;
        const float gx = disp[v * 3 + 0] + id_coord(x, W), gy = disp[v * 3 + 1] + id_coord(y, H), gz = disp[v * 3 + 2] + id_coord(z, D);
        if (!is_finite_coord(gx, gy, gz)) continue;
        const Taps t = make_taps(gx, gy, gz, D, H, W);
        const int lab = warp_label_at(lab_t, bt, v);
        const bool lok = lab >= 0 && lab < C;
        float* plane = lok ? B + ((long long)n * C + lab) * V : (A_extra ? A_extra + sbase : nullptr);
        if (!plane) continue;
        int slot = -1;
        if (lok) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (slot < 0) {
                    int cur = slot_label[k];
                    if (cur == -1) cur = atomicCAS(&slot_label[k], -1, lab), cur = cur == -1 ? lab : cur;
                    if (cur == lab) slot = k;
                }
            }
        }
        const int cx0 = t.x0 - ox, cy0 = t.y0 - oy, cz0 = t.z0 - oz;
        const bool inbox = slot >= 0 && cx0 >= 0 && cx0 + 1 < LX && cy0 >= 0 && cy0 + 1 < LY && cz0 >= 0 && cz0 + 1 < LZ;
        float* lb = box + slot * CELLS + (cz0 * LY + cy0) * LX + cx0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cz = k >> 2, cy = (k >> 1) & 1, cx = k & 1;
            const int xx = t.x0 + cx, yy = t.y0 + cy, zz = t.z0 + cz;
            const float wgt = (cx ? t.fx0 : t.fx1) * (cy ? t.fy0 : t.fy1) * (cz ? t.fz0 : t.fz1);
            if (xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D && wgt != 0.f) {
                if (inbox) atomicAdd(lb + (cz * LY + cy) * LX + cx, wgt);
                else atomicAdd(plane + ((long long)zz * H + yy) * W + xx, wgt);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * CELLS; i += 256) {
        const float val = box[i];
        if (val != 0.f) {
            const int k = i / CELLS, c = i - k * CELLS;
            const int cx = c % LX, cy = (c / LX) % LY, cz = c / (LX * LY);
            atomicAdd(B + ((long long)n * C + slot_label[k]) * V + ((long long)(oz + cz) * H + (oy + cy)) * W + (ox + cx), val);      // (only in-volume cells were added to)
        }
    }
}

// dlogits[u][j] = p[u][j] (g[u][j] - sum_c g[u][c] p[u][c]),  g = gl_a (b_a[c] A[u] + a_a[c] B[c][u]) + gl_s (a_s[c] [Sm[u] == c] + b_s[c]),
// p = softmax(logits) given as `prob` ([N][V][C]); B class-major ([N][C][V], see above); dlogits [N][V][C].  A workgroup takes `tv`
// consecutive voxels: the C planes' segments are read coalesced into an LDS tile [C][tv + 1], then lpv = C / 4 lanes per voxel form the
// row (16-byte accesses on prob / dlogits).
__global__ void __launch_bounds__(256) seg_anat_dlogits_kernel(const float* __restrict__ prob, const void* __restrict__ lab_m, int bm,
                                                               const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ dlogits,
                                                               const float* __restrict__ coef_s, const float* __restrict__ coef_a,
                                                               const float* __restrict__ gl_s, const float* __restrict__ gl_a,
                                                               int N, long long V, int C, int lpv, int tv) {
    extern __shared__ float tile[];                                        // [C][tv + 1]
    const int NC = N * C, ts = tv + 1;
    const int ltv = __ffs(tv) - 1, llp = __ffs(lpv) - 1;                   // tv, lpv: powers of two
    const float gs = (coef_s && gl_s) ? gl_s[0] : 0.f, ga = gl_a ? gl_a[0] : 0.f;
    const long long tps = (V + tv - 1) / tv, ntiles = tps * N;
    for (DaXcdItems TL = da_xcd_items(ntiles); TL.i < TL.end; TL.i += TL.step) {
        const long long tix = TL.i;
        const int n = (int)(tix / tps);
        const long long u0 = (tix - (long long)n * tps) * tv;
        __syncthreads();                                                   // the previous tile has been consumed
        const float* Bn = B + (long long)n * C * V;
        for (int idx = threadIdx.x; idx < C * tv; idx += 256) {
            const int c = idx >> ltv, t = idx & (tv - 1);
            const long long u = u0 + t;
            tile[c * ts + t] = (u < V) ? Bn[(long long)c * V + u] : 0.f;
        }
        __syncthreads();
        for (int item = threadIdx.x; item < tv * lpv; item += 256) {       // (256 % lpv == 0: the lanes of a voxel share a wave and a trip)
            const int t = item >> llp, q = item & (lpv - 1);
            const long long u = u0 + t;
            if (u >= V) continue;                                          // whole voxel groups drop out together
            const long long row = (long long)n * V + u;
            const float4 pv = *reinterpret_cast<const float4*>(prob + row * C + q * 4);
            const float b[4] = {tile[(q * 4 + 0) * ts + t], tile[(q * 4 + 1) * ts + t], tile[(q * 4 + 2) * ts + t], tile[(q * 4 + 3) * ts + t]};
            float a = b[0] + b[1] + b[2] + b[3];                           // A[u] = sum_c B[c][u] (+ the out-of-range-label weights)
            for (int k = 1; k < lpv; k <<= 1) a += __shfl_xor(a, k);
            if (A) a += A[row];
            const float p[4] = {pv.x, pv.y, pv.z, pv.w};
            const int lab = (coef_s && lab_m) ? warp_label_at(lab_m, bm, row) - q * 4 : -1;
            float g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = n * C + q * 4 + j;
                g[j] = ga * (coef_a[NC + c] * a + coef_a[c] * b[j]);
                if (coef_s) g[j] += gs * (coef_s[c] * (lab == j ? 1.f : 0.f) + coef_s[NC + c]);
            }
            float dot = g[0] * p[0] + g[1] * p[1] + g[2] * p[2] + g[3] * p[3];
            for (int k = 1; k < lpv; k <<= 1) dot += __shfl_xor(dot, k);
            *reinterpret_cast<float4*>(dlogits + row * C + q * 4) = make_float4(p[0] * (g[0] - dot), p[1] * (g[1] - dot), p[2] * (g[2] - dot), p[3] * (g[3] - dot));
        }
    }
}

// The same pass with ONE lane per voxel (C a compile-time constant <= 32): the C class planes are read directly -- lane = voxel, so every
// plane access is coalesced and needs no LDS tile, no shuffle and no barrier -- and the lane walks its own 128-byte rows of prob / dlogits
// in 16-byte pieces (eight accesses to the same line: the first one brings it into the CU's cache).
template <int CC>
__global__ void __launch_bounds__(256) seg_anat_dlogits_lane_kernel(const float* __restrict__ prob, const void* __restrict__ lab_m, int bm,
                                                                    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ dlogits,
                                                                    const float* __restrict__ coef_s, const float* __restrict__ coef_a,
                                                                    const float* __restrict__ gl_s, const float* __restrict__ gl_a, int N, long long V) {
    const int NC = N * CC;
    const float gs = (coef_s && gl_s) ? gl_s[0] : 0.f, ga = gl_a ? gl_a[0] : 0.f;
    const long long bps = (V + 255) / 256, nb = bps * N;                   // a workgroup never straddles two samples: n is uniform
    for (DaXcdItems BL = da_xcd_items(nb); BL.i < BL.end; BL.i += BL.step) {
        const long long bix = BL.i;
        const int n = (int)(bix / bps);
        const long long u = (bix - (long long)n * bps) * 256 + threadIdx.x;
        if (u >= V) continue;
        const long long row = (long long)n * V + u;
        const float* Bn = B + (long long)n * CC * V + u;
        float b[CC], p[CC];
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < CC; ++c) { b[c] = Bn[(long long)c * V]; a += b[c]; }
        if (A) a += A[row];
#pragma unroll
        for (int q = 0; q < CC / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(prob + row * CC + q * 4);
            p[4 * q] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
        }
        const int lab = (coef_s && lab_m) ? warp_label_at(lab_m, bm, row) : -1;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            float g = ga * (coef_a[NC + n * CC + c] * a + coef_a[n * CC + c] * b[c]);
            if (coef_s) g += gs * (coef_s[n * CC + c] * (lab == c ? 1.f : 0.f) + coef_s[NC + n * CC + c]);
            b[c] = g;
            dot += g * p[c];
        }
#pragma unroll
        for (int q = 0; q < CC / 4; ++q)
            *reinterpret_cast<float4*>(dlogits + row * CC + q * 4) = make_float4(p[4 * q] * (b[4 * q] - dot), p[4 * q + 1] * (b[4 * q + 1] - dot),
                                                                                 p[4 * q + 2] * (b[4 * q + 2] - dot), p[4 * q + 3] * (b[4 * q + 3] - dot));
    }
}

static bool vec_ok(int C, int* lpv) {
    if (C % 4 != 0) { *lpv = 1; return false; }
    const int q = C / 4;
    if (q > 64 || (q & (q - 1)) != 0) { *lpv = 1; return false; }
    *lpv = q;
    return true;
}

}  // namespace

extern "C" int da_warp_fwd(const float* src, const float* disp, float* deform, float* out,
                           int N, int D, int H, int W, int C, void* stream) {
    if (!src || !disp || !out || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0) return DA_ERR_BADARG;
    int lpv; const bool v4 = vec_ok(C, &lpv);
    const long long total = (long long)N * D * H * W * lpv;
    static const bool grouped = [] { const char* e = getenv("DA_WARP_GROUPED"); return !(e && atoi(e) == 0); }();      // A/B: 0 = the per-lane-group kernels
    if (v4 && grouped && lpv >= 2 && gather_grouped_ok(D, H, W, C, lpv) && N <= 65535) {
        int nb = (int)da_cdiv((long long)D * H * W, 512); if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
        hipLaunchKernelGGL(warp_fwd_grouped_kernel, dim3(nb, N), dim3(256), (size_t)4 * 64 * lpv * sizeof(float4), da_stream(stream), src, disp, deform, out, D, H, W, C, lpv);
    } else if (v4) hipLaunchKernelGGL((warp_fwd_kernel<4>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), src, disp, deform, out, N, D, H, W, C, lpv);
    else hipLaunchKernelGGL((warp_fwd_kernel<1>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), src, disp, deform, out, N, D, H, W, C, lpv);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_warp_bwd(const float* dout, const float* src, const float* disp, float* d_disp, float* d_src,
                           int N, int D, int H, int W, int C, void* stream) {
    if (!dout || !src || !disp || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0) return DA_ERR_BADARG;
    if (!d_disp && !d_src) return 0;
    int lpv; const bool v4 = vec_ok(C, &lpv);
    if (d_src && C >= 8 && C <= 64 && 64 % C == 0) {
        const long long tot = (long long)N * D * H * W * C;
        hipLaunchKernelGGL(warp_bwd_dsrc_lane_kernel, dim3(da_grid(tot, 256)), dim3(256), 0, da_stream(stream), dout, disp, d_src, N, D, H, W, C);
        d_src = nullptr;
        if (!d_disp) { DA_LAUNCH_CHECK(); return 0; }
    }
    const long long total = (long long)N * D * H * W * lpv;
    // block = 256 and gridDim*256 are multiples of lpv (<= 64), so the lanes of one voxel share a wave and a trip count
    if (v4) hipLaunchKernelGGL((warp_bwd_kernel<4>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), dout, src, disp, d_disp, d_src, N, D, H, W, C, lpv);
    else hipLaunchKernelGGL((warp_bwd_kernel<1>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), dout, src, disp, d_disp, d_src, N, D, H, W, C, lpv);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_identity_grid(float* out, int D, int H, int W, int normalize, void* stream) {
    if (!out || D < 1 || H < 1 || W < 1) return DA_ERR_BADARG;
    const long long V = (long long)D * H * W;
    hipLaunchKernelGGL(identity_grid_kernel, dim3(da_grid(V, 256)), dim3(256), 0, da_stream(stream), out, D, H, W, normalize);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_warp_labels_fwd(const void* labels, int label_bytes, const float* disp, float* out,
                                  int N, int D, int H, int W, int C, void* stream) {
    if (!labels || !disp || !out || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0 || (label_bytes != 1 && label_bytes != 8)) return DA_ERR_BADARG;
    const long long total = (long long)N * D * H * W * C;
    if (C % 4 == 0) hipLaunchKernelGGL((warp_labels_fwd_kernel<4>), dim3(da_grid(total / 4, 256)), dim3(256), 0, da_stream(stream), labels, label_bytes, disp, out, N, D, H, W, C);
    else hipLaunchKernelGGL((warp_labels_fwd_kernel<1>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), labels, label_bytes, disp, out, N, D, H, W, C);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_warp_labels_bwd(const float* dout, const void* labels, int label_bytes, const float* disp, float* d_disp,
                                  int N, int D, int H, int W, int C, void* stream) {
    if (!dout || !labels || !disp || !d_disp || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0 || (label_bytes != 1 && label_bytes != 8)) return DA_ERR_BADARG;
    const long long nvox = (long long)N * D * H * W;
    hipLaunchKernelGGL(warp_labels_bwd_kernel, dim3(da_grid(nvox, 256)), dim3(256), 0, da_stream(stream), dout, labels, label_bytes, disp, d_disp, N, D, H, W, C);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t da_warp_bwd_dsrc_det_ws_bytes(int N, int D, int H, int W, int C) {
    return da_align((size_t)N * D * H * W * C * sizeof(unsigned long long)) + 256;
}

extern "C" int da_warp_bwd_dsrc_det(const float* dout, const float* disp, float* d_src, int N, int D, int H, int W, int C,
                                    void* ws, size_t ws_bytes, void* stream) {
    if (!dout || !disp || !d_src || !ws || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0) return DA_ERR_BADARG;
    if (ws_bytes < da_warp_bwd_dsrc_det_ws_bytes(N, D, H, W, C)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    const long long tot = (long long)N * D * H * W * C;
    const size_t acc_bytes = da_align((size_t)tot * sizeof(unsigned long long));
    unsigned long long* acc = (unsigned long long*)ws;
    unsigned int* max_bits = (unsigned int*)((char*)ws + acc_bytes);
    hipError_t e = hipMemsetAsync(ws, 0, acc_bytes + 256, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(absmax_kernel, dim3(da_grid(tot, 256)), dim3(256), 0, st, dout, tot, max_bits);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(warp_bwd_dsrc_fixed_kernel, dim3(da_grid(tot, 256)), dim3(256), 0, st, dout, disp, acc, (const unsigned int*)max_bits, N, D, H, W, C);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(fixed_to_float_kernel, dim3(da_grid(tot, 256)), dim3(256), 0, st, (const unsigned long long*)acc, (const unsigned int*)max_bits, d_src, tot);
    DA_LAUNCH_CHECK();
    return 0;
}

// ---- fused anatomy losses (joint step) -------------------------------------------------------------------------------------
static const int kLwdBlocks = 4096;          // (a gather kernel hides its two dependent memory latencies with waves: 16 workgroups per CU)

extern "C" size_t da_label_warp_dice_ws_bytes(int N, int C) {
    return da_align((size_t)N * kLwdBlocks * 3 * C * sizeof(double)) + da_align((size_t)3 * N * C * sizeof(float));
}

extern "C" int da_label_warp_dice_fwd(const void* lab_m, int lab_m_bytes, const void* lab_t, int lab_t_bytes, const float* disp,
                                      int N, int D, int H, int W, int C, int weight_type, int no_bg, float eps,
                                      float* loss, float* coef, void* ws, size_t ws_bytes, void* stream) {
    if (!lab_m || !lab_t || !disp || !loss || !coef || N <= 0 || N > 64 || D < 2 || H < 2 || W < 2 || C <= 0 ||
        (lab_m_bytes != 1 && lab_m_bytes != 8) || (lab_t_bytes != 1 && lab_t_bytes != 8)) return DA_ERR_BADARG;
    if (C > 64) return DA_ERR_UNSUPPORTED;                               // callers fall back to warp(one-hot) + Dice
    if (ws_bytes < da_label_warp_dice_ws_bytes(N, C)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    double* partial = (double*)ws;
    float* isc = (float*)((char*)ws + da_align((size_t)N * kLwdBlocks * 3 * C * sizeof(double)));
    const long long V = (long long)D * H * W;
    int nblocks = (int)da_cdiv(V, 256 * 2); if (nblocks > kLwdBlocks) nblocks = kLwdBlocks; if (nblocks < 1) nblocks = 1;
    hipLaunchKernelGGL(label_warp_dice_partial_kernel, dim3(nblocks, N), dim3(256), 0, st, lab_m, lab_m_bytes, lab_t, lab_t_bytes, disp, D, H, W, C, partial);
    DA_LAUNCH_CHECK();
    return da_dice_finish(partial, nblocks, N, C, weight_type, no_bg, eps, loss, coef, isc, st);
}

extern "C" size_t da_warp_dice_ws_bytes(int N, int C) { return da_label_warp_dice_ws_bytes(N, C); }

/* loss = Dice(warp(src, identity + disp), onehot(lab_t)) (src = probabilities, softmax = 0 in da_dice_fwd's terms) and its backward
 * coefficients, the warped tensor never written: what da_warp_fwd + da_dice_fwd compute, minus one write and one read of N V C floats */
extern "C" int da_warp_dice_fwd(const float* src, const float* disp, const void* lab_t, int lab_t_bytes,
                                int N, int D, int H, int W, int C, int weight_type, int no_bg, float eps,
                                float* loss, float* coef, void* ws, size_t ws_bytes, void* stream) {
    if (!src || !disp || !lab_t || !loss || !coef || N <= 0 || N > 64 || D < 2 || H < 2 || W < 2 || C <= 0 ||
        (lab_t_bytes != 1 && lab_t_bytes != 8)) return DA_ERR_BADARG;
    int lpv; if (!vec_ok(C, &lpv) || C > 64) return DA_ERR_UNSUPPORTED;       // callers run da_warp_fwd + da_dice_fwd
    if (ws_bytes < da_warp_dice_ws_bytes(N, C)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    double* partial = (double*)ws;
    float* isc = (float*)((char*)ws + da_align((size_t)N * kLwdBlocks * 3 * C * sizeof(double)));
    const long long V = (long long)D * H * W;
    const int slots = 256 / lpv;
    int nblocks = (int)da_cdiv(V, (long long)slots * 4); if (nblocks > kLwdBlocks) nblocks = kLwdBlocks; if (nblocks < 1) nblocks = 1;
    static const bool grouped = [] { const char* e = getenv("DA_WARP_GROUPED"); return !(e && atoi(e) == 0); }();
    if (grouped && lpv >= 2 && gather_grouped_ok(D, H, W, C, lpv))
        hipLaunchKernelGGL(warp_dice_grouped_kernel, dim3(nblocks, N), dim3(256), (size_t)3 * slots * C * sizeof(float), st,
                           src, disp, lab_t, lab_t_bytes, D, H, W, C, lpv, partial);
    else
        hipLaunchKernelGGL(warp_dice_partial_kernel, dim3(nblocks, N), dim3(256), (size_t)3 * slots * C * sizeof(float), st,
                           src, disp, lab_t, lab_t_bytes, D, H, W, C, lpv, partial);
    DA_LAUNCH_CHECK();
    return da_dice_finish(partial, nblocks, N, C, weight_type, no_bg, eps, loss, coef, isc, st);
}

extern "C" int da_label_warp_dice_bwd(const void* lab_m, int lab_m_bytes, const void* lab_t, int lab_t_bytes, const float* disp,
                                      const float* coef, const float* dloss, float* d_disp, int N, int D, int H, int W, int C, void* stream) {
    if (!lab_m || !lab_t || !disp || !coef || !dloss || !d_disp || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0 ||
        (lab_m_bytes != 1 && lab_m_bytes != 8) || (lab_t_bytes != 1 && lab_t_bytes != 8)) return DA_ERR_BADARG;
    const long long nvox = (long long)N * D * H * W;
    hipLaunchKernelGGL(label_warp_dice_bwd_kernel, dim3(da_grid(nvox, 256)), dim3(256), 0, da_stream(stream), lab_m, lab_m_bytes, lab_t, lab_t_bytes,
                       disp, coef, dloss, d_disp, N, D, H, W, C);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_warp_adjoint_labels(const void* lab_t, int lab_t_bytes, const float* disp, float* A, float* B,
                                      int N, int D, int H, int W, int C, void* stream) {
    if (!lab_t || !disp || !B || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0 || (lab_t_bytes != 1 && lab_t_bytes != 8)) return DA_ERR_BADARG;
    hipStream_t st = da_stream(stream);
    const long long nvox = (long long)N * D * H * W;
    hipError_t e = hipMemsetAsync(B, 0, (size_t)nvox * C * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (A) { e = hipMemsetAsync(A, 0, (size_t)nvox * sizeof(float), st); if (e != hipSuccess) return (int)e; }
    static const int use_box = [] { const char* e = getenv("DA_ADJ_BOX"); return (e && e[0] == '0') ? 0 : 1; }();
    if (use_box) {
        constexpr int BX = 32, BY = 8, BZ = 4, M = 2, K = 3;
        constexpr size_t shm = (size_t)K * (BX + 2 * M + 1) * (BY + 2 * M + 1) * (BZ + 2 * M + 1) * sizeof(float);
        auto kern = warp_adjoint_labels_box_kernel<BX, BY, BZ, M, K>;
        static bool attr_set = false;
        if (!attr_set) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        const int nbx = (W + BX - 1) / BX, nby = (H + BY - 1) / BY, nbz = (D + BZ - 1) / BZ;
        const long long nb = (long long)nbx * nby * nbz * N;
        if (nb < (1ll << 31)) {
            hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), shm, st, lab_t, lab_t_bytes, disp, A, B, N, D, H, W, C, nbx, nby, nbz);
            DA_LAUNCH_CHECK();
            return 0;
        }
    }
    hipLaunchKernelGGL(warp_adjoint_labels_kernel, dim3(da_grid(nvox, 256)), dim3(256), 0, st, lab_t, lab_t_bytes, disp, A, B, N, D, H, W, C);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_seg_anat_dlogits(const float* prob, const void* lab_m, int lab_m_bytes, const float* A, const float* B, float* dlogits,
                                   const float* coef_sup, const float* coef_anat, const float* dloss_sup, const float* dloss_anat,
                                   int N, long long V, int C, void* stream) {
    if (!prob || !B || !dlogits || !coef_anat || !dloss_anat || N <= 0 || V <= 0 || C <= 0) return DA_ERR_BADARG;
    if (coef_sup && (!lab_m || !dloss_sup || (lab_m_bytes != 1 && lab_m_bytes != 8))) return DA_ERR_BADARG;
    int lpv; if (!vec_ok(C, &lpv)) return DA_ERR_UNSUPPORTED;
    if (C == 32 && !getenv("DA_DLOGITS_TILE")) {
        const long long nb = ((V + 255) / 256) * N;
        hipLaunchKernelGGL((seg_anat_dlogits_lane_kernel<32>), dim3(da_grid(nb * 256, 256)), dim3(256), 0, da_stream(stream), prob, lab_m, lab_m_bytes, A, B, dlogits,
                           coef_sup, coef_anat, dloss_sup, dloss_anat, N, V);
        DA_LAUNCH_CHECK();
        return 0;
    }
    const int tv = C <= 32 ? 256 : (C <= 64 ? 128 : 32);                  // LDS tile [C][tv + 1] floats <= 33 KB
    const long long ntiles = ((V + tv - 1) / tv) * N;
    const size_t shm = (size_t)C * (tv + 1) * sizeof(float);
    hipLaunchKernelGGL(seg_anat_dlogits_kernel, dim3(da_grid(ntiles * 256, 256)), dim3(256), shm, da_stream(stream), prob, lab_m, lab_m_bytes, A, B, dlogits,
                       coef_sup, coef_anat, dloss_sup, dloss_anat, N, V, C, lpv, tv);
    DA_LAUNCH_CHECK();
    return 0;
}
