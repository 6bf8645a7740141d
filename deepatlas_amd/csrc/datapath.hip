// Device data path: the tensor halves of lib/transforms.py -- SitkToTensor's clamp/cast (:71-92), CropTensor
// (:124-158), Partition's overlap tiling with reflect padding and its two assemble modes (:508-649).  Plain HBM-bound copy kernels:
// one thread per output element, 64-bit indexing, no intermediate padded volume (the reflect index is computed per element).
#include "common.h"

namespace {

__device__ __forceinline__ float load_as_f32(const void* src, int dtype, long long i) {
    switch (dtype) {
        case 0: return ((const float*)src)[i];
        case 1: return (float)((const double*)src)[i];
        case 2: return (float)((const short*)src)[i];
        case 3: return (float)((const unsigned char*)src)[i];
        default: return (float)((const int*)src)[i];
    }
}

// img_np[img_np > 1] = 1; img_np[img_np < 0] = 0; np.float32(img_np)   (comparison in the source dtype, then the cast)
__global__ void clamp01_to_f32_kernel(const void* __restrict__ src, int dtype, float* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v;
        if (dtype == 1) { double d = ((const double*)src)[i]; d = d > 1.0 ? 1.0 : d; d = d < 0.0 ? 0.0 : d; v = (float)d; }
        else { v = load_as_f32(src, dtype, i); v = v > 1.f ? 1.f : v; v = v < 0.f ? 0.f : v; }
        dst[i] = v;
    }
}

template <typename T>
__global__ void crop3d_kernel(const T* __restrict__ src, T* __restrict__ dst, long long C, int D, int H, int W,
                              int d0, int h0, int w0, int Do, int Ho, int Wo) {
    const long long total = C * Do * Ho * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(i % Wo); long long r = i / Wo;
        const int h = (int)(r % Ho); r /= Ho;
        const int d = (int)(r % Do); const long long c = r / Do;
        dst[i] = src[((c * D + d0 + d) * H + h0 + h) * W + w0 + w];
    }
}

// numpy.pad(mode='reflect') index: mirror without repeating the edge, applied repeatedly for pads wider than the axis
__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    int m = i % period; if (m < 0) m += period;
    return m < n ? m : period - m;
}

struct TileGeom { int D, H, W; int tz, ty, tx; int oz, oy, ox; int gz, gy, gx; };   // volume, tile, overlap, tile grid (numpy order z, y, x)

// tiles[(i*gy + j)*gx + k][z][y][x] = padded[i*ez + z][j*ey + y][k*ex + x], padded = reflect-pad by `overlap` in front
template <typename T>
__global__ void partition_kernel(const T* __restrict__ vol, T* __restrict__ tiles, TileGeom g) {
    const int ez = g.tz - 2 * g.oz, ey = g.ty - 2 * g.oy, ex = g.tx - 2 * g.ox;
    const long long per = (long long)g.tz * g.ty * g.tx;
    const long long total = per * g.gz * g.gy * g.gx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % g.tx); long long r = i / g.tx;
        const int y = (int)(r % g.ty); r /= g.ty;
        const int z = (int)(r % g.tz); long long t = r / g.tz;
        const int k = (int)(t % g.gx); t /= g.gx;
        const int j = (int)(t % g.gy); const int ii = (int)(t / g.gy);
        const int sz = reflect_idx(ii * ez + z - g.oz, g.D), sy = reflect_idx(j * ey + y - g.oy, g.H), sx = reflect_idx(k * ex + x - g.ox, g.W);
        tiles[i] = vol[((long long)sz * g.H + sy) * g.W + sx];
    }
}

// non-voting assemble: every voxel comes from the effective (un-overlapped) core of exactly one tile
template <typename T>
__global__ void assemble_kernel(const T* __restrict__ tiles, T* __restrict__ vol, TileGeom g) {
    const int ez = g.tz - 2 * g.oz, ey = g.ty - 2 * g.oy, ex = g.tx - 2 * g.ox;
    const long long total = (long long)g.D * g.H * g.W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % g.W); long long r = i / g.W;
        const int y = (int)(r % g.H); const int z = (int)(r / g.H);
        const int ti = z / ez, tj = y / ey, tk = x / ex;
        const long long t = ((long long)ti * g.gy + tj) * g.gx + tk;
        vol[i] = tiles[((t * g.tz + (z - ti * ez + g.oz)) * g.ty + (y - tj * ey + g.oy)) * g.tx + (x - tk * ex + g.ox)];
    }
}

// voting assemble: every tile covering the (padded-frame) voxel votes with its label; the label with most votes wins, ties go to
// the smallest label (np.argmax over the label axis).  Labels are < 256; at most MAXC covering tiles per axis.
__global__ void assemble_vote_kernel(const unsigned char* __restrict__ tiles, unsigned char* __restrict__ vol, TileGeom g) {
    const int ez = g.tz - 2 * g.oz, ey = g.ty - 2 * g.oy, ex = g.tx - 2 * g.ox;
    const long long total = (long long)g.D * g.H * g.W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % g.W); long long r = i / g.W;
        const int y = (int)(r % g.H); const int z = (int)(r / g.H);
        const int pz = z + g.oz, py = y + g.oy, px = x + g.ox;            // coordinates in the padded frame
        unsigned char labs[64]; int nl = 0;
        // tile index a covers padded coordinate p iff a*e <= p < a*e + tile
        for (int a = max(0, (pz - g.tz) / ez); a < g.gz && a * ez <= pz; ++a) {
            if (pz >= a * ez + g.tz) continue;
            for (int b = max(0, (py - g.ty) / ey); b < g.gy && b * ey <= py; ++b) {
                if (py >= b * ey + g.ty) continue;
                for (int c = max(0, (px - g.tx) / ex); c < g.gx && c * ex <= px; ++c) {
                    if (px >= c * ex + g.tx) continue;
                    const long long t = ((long long)a * g.gy + b) * g.gx + c;
                    if (nl < 64) labs[nl++] = tiles[((t * g.tz + (pz - a * ez)) * g.ty + (py - b * ey)) * g.tx + (px - c * ex)];
                }
            }
        }
        int best = 0, bestc = -1;
        for (int a = 0; a < nl; ++a) {
            int cnt = 0;
            for (int b = 0; b < nl; ++b) cnt += (labs[b] == labs[a]);
            if (cnt > bestc || (cnt == bestc && labs[a] < best)) { bestc = cnt; best = labs[a]; }
        }
        vol[i] = (unsigned char)best;
    }
}

// ---- synthetic volumes (SURVEY.md row f4 "synthetic-volume generator on device"; BASELINE configs are all synthetic) ----------
// Counter-based: every element is a pure function of (seed, sample, voxel index), so the numpy restatement (oracle/datapath.py
// synth_volume) reproduces it bit for bit and ranks / batch sizes do not change what a given sample looks like.
__device__ __forceinline__ unsigned int mix32(unsigned int x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned int synth_bits(unsigned int seed, unsigned long long i, unsigned int stream_id) {
    const unsigned int hi = (unsigned int)(i >> 32), lo = (unsigned int)i;
    return mix32(mix32(lo ^ mix32(seed + 0x9e3779b9u * hi)) + stream_id);
}

// mode 0 (throughput inputs): img ~ U[0,1), lab ~ U{0..C-1}, iid.
// mode 1 (Dice-parity inputs): blocky label map lab = ((z / bz) * 5 + (y / by) * 3 + x / bx + sample) % C with b = max(dim / 8, 1),
//         img = clamp(lab / (C - 1) + noise * U, 0, 1)   -- the structure lib/datasets.py's synthetic dataset uses.
__global__ void synth_volume_kernel(float* __restrict__ img, unsigned char* __restrict__ lab, int N, int D, int H, int W, int C,
                                    int mode, float noise, unsigned int seed, int sample0) {
    const long long V = (long long)D * H * W, total = V * N;
    const int bz = max(D / 8, 1), by = max(H / 8, 1), bx = max(W / 8, 1);
    const float inv24 = 1.0f / 16777216.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / V, v = i - n * V;
        const unsigned long long ctr = (unsigned long long)(sample0 + n) * (unsigned long long)V + (unsigned long long)v;
        const float u = (float)(synth_bits(seed, ctr, 0u) >> 8) * inv24;
        if (mode == 0) {
            if (img) img[i] = u;
            if (lab) lab[i] = (unsigned char)(synth_bits(seed, ctr, 1u) % (unsigned int)C);
        } else {
            const int x = (int)(v % W); const long long r = v / W;
            const int y = (int)(r % H), z = (int)(r / H);
            const int l = (int)(((long long)(z / bz) * 5 + (y / by) * 3 + (x / bx) + sample0 + n) % C);
            if (lab) lab[i] = (unsigned char)l;
            if (img) {
                float val;
                {
#pragma clang fp contract(off)                                   // numpy order: round the quotient, the product and the sum separately
                    const float q = (float)l / (float)max(C - 1, 1);
                    const float pn = noise * u;
                    val = q + pn;
                }
                img[i] = fminf(fmaxf(val, 0.f), 1.f);
            }
        }
    }
}

}  // namespace

extern "C" int da_clamp01_to_f32(const void* src, int src_dtype, float* dst, long long n, void* stream) {
    if (!src || !dst || n <= 0 || src_dtype < 0 || src_dtype > 4) return DA_ERR_BADARG;
    hipLaunchKernelGGL(clamp01_to_f32_kernel, dim3(da_grid(n, 256)), dim3(256), 0, da_stream(stream), src, src_dtype, dst, n);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_crop3d(const void* src, void* dst, int elem_bytes, long long C, int D, int H, int W,
                         int d0, int h0, int w0, int Do, int Ho, int Wo, void* stream) {
    if (!src || !dst || C <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || d0 < 0 || h0 < 0 || w0 < 0 || d0 + Do > D || h0 + Ho > H || w0 + Wo > W) return DA_ERR_BADARG;
    const long long total = C * Do * Ho * Wo;
    if (elem_bytes == 4) hipLaunchKernelGGL((crop3d_kernel<float>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), (const float*)src, (float*)dst, C, D, H, W, d0, h0, w0, Do, Ho, Wo);
    else if (elem_bytes == 1) hipLaunchKernelGGL((crop3d_kernel<unsigned char>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), (const unsigned char*)src, (unsigned char*)dst, C, D, H, W, d0, h0, w0, Do, Ho, Wo);
    else return DA_ERR_BADARG;
    DA_LAUNCH_CHECK();
    return 0;
}

static int tile_geom(TileGeom& g, int D, int H, int W, const int* tile3, const int* overlap3) {
    g.D = D; g.H = H; g.W = W; g.tz = tile3[0]; g.ty = tile3[1]; g.tx = tile3[2]; g.oz = overlap3[0]; g.oy = overlap3[1]; g.ox = overlap3[2];
    const int ez = g.tz - 2 * g.oz, ey = g.ty - 2 * g.oy, ex = g.tx - 2 * g.ox;
    if (D <= 0 || H <= 0 || W <= 0 || ez <= 0 || ey <= 0 || ex <= 0 || g.oz < 0 || g.oy < 0 || g.ox < 0) return DA_ERR_BADARG;
    g.gz = (D + ez - 1) / ez; g.gy = (H + ey - 1) / ey; g.gx = (W + ex - 1) / ex;
    return 0;
}

// tile3 / overlap3 in numpy order (z, y, x); tiles: [gz*gy*gx][tz][ty][tx]
extern "C" int da_partition_tiles(const void* vol, void* tiles, int elem_bytes, int D, int H, int W, const int* tile3, const int* overlap3, void* stream) {
    TileGeom g;
    if (!vol || !tiles || !tile3 || !overlap3 || tile_geom(g, D, H, W, tile3, overlap3)) return DA_ERR_BADARG;
    const long long total = (long long)g.tz * g.ty * g.tx * g.gz * g.gy * g.gx;
    if (elem_bytes == 4) hipLaunchKernelGGL((partition_kernel<float>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), (const float*)vol, (float*)tiles, g);
    else if (elem_bytes == 1) hipLaunchKernelGGL((partition_kernel<unsigned char>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), (const unsigned char*)vol, (unsigned char*)tiles, g);
    else return DA_ERR_BADARG;
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_assemble_tiles(const void* tiles, void* vol, int elem_bytes, int D, int H, int W, const int* tile3, const int* overlap3, int vote, void* stream) {
    TileGeom g;
    if (!vol || !tiles || !tile3 || !overlap3 || tile_geom(g, D, H, W, tile3, overlap3)) return DA_ERR_BADARG;
    const long long total = (long long)D * H * W;
    if (vote) {
        if (elem_bytes != 1) return DA_ERR_BADARG;
        const int ez = g.tz - 2 * g.oz, ey = g.ty - 2 * g.oy, ex = g.tx - 2 * g.ox;
        if ((long long)((g.tz + ez - 1) / ez) * ((g.ty + ey - 1) / ey) * ((g.tx + ex - 1) / ex) > 64) return DA_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(assemble_vote_kernel, dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), (const unsigned char*)tiles, (unsigned char*)vol, g);
    } else if (elem_bytes == 4) hipLaunchKernelGGL((assemble_kernel<float>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), (const float*)tiles, (float*)vol, g);
    else if (elem_bytes == 1) hipLaunchKernelGGL((assemble_kernel<unsigned char>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), (const unsigned char*)tiles, (unsigned char*)vol, g);
    else return DA_ERR_BADARG;
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_synth_volume(float* img, unsigned char* labels, int N, int D, int H, int W, int n_classes, int mode, float noise,
                               unsigned int seed, int sample0, void* stream) {
    if ((!img && !labels) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || n_classes < 1 || n_classes > 256 || (mode != 0 && mode != 1) || sample0 < 0) return DA_ERR_BADARG;
    const long long total = (long long)N * D * H * W;
    hipLaunchKernelGGL(synth_volume_kernel, dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), img, labels, N, D, H, W, n_classes, mode, noise, seed, sample0);
    DA_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// bf16 activation storage (common.h): conversions at the network boundary and around kernels that have no bf16 twin for a shape.
// fp32 -> bf16 rounds to nearest-even (the one rounding a stored activation gets); bf16 -> fp32 is exact.
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ void cast_f32_to_bf16_kernel(const float* __restrict__ x, da_bf16* __restrict__ y, long long n) {
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) da_stq(y, i, da_ldq(x, i));
    for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) da_st1(y, i, x[i]);
}
__global__ void cast_bf16_to_f32_kernel(const da_bf16* __restrict__ x, float* __restrict__ y, long long n) {
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) da_stq(y, i, da_ldq(x, i));
    for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = da_ld1(x, i);
}
}  // namespace

extern "C" int da_cast_f32_to_bf16(const float* x, void* y, long long numel, void* stream) {
    if (!x || !y || numel <= 0) return DA_ERR_BADARG;
    hipLaunchKernelGGL(cast_f32_to_bf16_kernel, dim3(da_grid(numel / 4 + 1, 256)), dim3(256), 0, da_stream(stream), x, (da_bf16*)y, numel);
    DA_LAUNCH_CHECK();
    return 0;
}
extern "C" int da_cast_bf16_to_f32(const void* x, float* y, long long numel, void* stream) {
    if (!x || !y || numel <= 0) return DA_ERR_BADARG;
    hipLaunchKernelGGL(cast_bf16_to_f32_kernel, dim3(da_grid(numel / 4 + 1, 256)), dim3(256), 0, da_stream(stream), (const da_bf16*)x, y, numel);
    DA_LAUNCH_CHECK();
    return 0;
}
