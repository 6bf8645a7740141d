// Deformation-field trilinear warp (spatial transformer) forward/backward + identity grid.
// Rows a9-a10 of SURVEY.md §8: `deform = disp + identity` (voxel_morph.py:85-88, lib/utils.py:89-102) and
// F.grid_sample(src, deform.permute(0,2,3,4,1), 'bilinear', 'zeros', align_corners=True) (voxel_morph.py:90-91).
// NDHWC: src[N][D][H][W][C], disp[N][D][H][W][3] with channel order (x, y, z) = (W, H, D) axis.
// HBM-bound gather: 12-byte disp read + 8 corner taps of C contiguous floats (16-byte lanes when C % 4 == 0).
#include "common.h"

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
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % lpv);
        const long long v = i / lpv;
        long long r = v;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int d = (int)(r % D); const int n = (int)(r / D);
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

template <int VEC>
__global__ void warp_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ src,
                                const float* __restrict__ disp, float* __restrict__ d_disp, float* __restrict__ d_src,
                                int N, int D, int H, int W, int C, int lpv) {
    const long long nvox = (long long)N * D * H * W;
    const long long total = nvox * lpv;
    // total is padded by the launcher to a multiple of lpv*...; every lane of a voxel group runs the same trip count
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % lpv);
        const long long v = i / lpv;
        long long r = v;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int d = (int)(r % D); const int n = (int)(r / D);
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
        const int c = (int)(i % C);
        const long long v = i / C;
        long long r = v;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int d = (int)(r % D); const int n = (int)(r / D);
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
        const int c0 = (int)(i % cq) * VEC;
        const long long v = i / cq;
        long long r = v;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int d = (int)(r % D); const int n = (int)(r / D);
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
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (long long)gridDim.x * blockDim.x) {
        long long r = v;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int d = (int)(r % D); const int n = (int)(r / D);
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
        long long r = i;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const int d = (int)(r / H);
        out[i] = normalize ? id_coord(w, W) : (float)w;           // channel 0: W axis (x)
        out[V + i] = normalize ? id_coord(h, H) : (float)h;       // channel 1: H axis (y)
        out[2 * V + i] = normalize ? id_coord(d, D) : (float)d;   // channel 2: D axis (z)
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
    if (v4) hipLaunchKernelGGL((warp_fwd_kernel<4>), dim3(da_grid(total, 256)), dim3(256), 0, da_stream(stream), src, disp, deform, out, N, D, H, W, C, lpv);
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
