// 2x2x2 stride-2 transposed convolution (row a3: nn.ConvTranspose3d(k=2,s=2), unets.py:49,55,240-241) and the
// 1x1x1 segmentation head (row a5: nn.Conv3d(dec[-1], n_classes, 1), unets.py:249-250).  NDHWC, fp32.
// Both are HBM-bound at DeepAtlas' widths (the up-sampled tensor is the largest activation): one input voxel
// per lane, weights through wave-uniform scalar loads, each lane writes contiguous runs of output channels.
#include "common.h"
#include "pointwise_internal.h"

namespace {

// One lane = one INPUT voxel and one (i, j) output row parity; it emits both k = 0, 1 outputs, i.e. a contiguous
// run of 2*Cout floats, for CT output channels at a time.
template <int CT>
__global__ void __launch_bounds__(256)
deconv_k2s2_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                       float* __restrict__ out, int N, int D, int H, int W, int Cin, int Cout) {
    const long long nvox = (long long)N * D * H * W;
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    const int co0 = blockIdx.y * CT;
    long long r = v;
    const int iw = (int)(r % W); r /= W;
    const int ih = (int)(r % H); r /= H;
    const int id = (int)(r % D); const int n = (int)(r / D);
    const float* x = in + v * Cin;
    const int Ho = 2 * H, Wo = 2 * W, Do = 2 * D;
#pragma unroll 1
    for (int ij = 0; ij < 4; ++ij) {
        const int i = ij >> 1, j = ij & 1;
        float acc[2][CT];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[k][c] = (bias && co0 + c < Cout) ? bias[co0 + c] : 0.f;
        const float* w0 = w + ((size_t)(ij * 2 + 0) * Cin) * Cout + co0;
        const float* w1 = w + ((size_t)(ij * 2 + 1) * Cin) * Cout + co0;
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = x[ci];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (co0 + c < Cout) {
                    acc[0][c] += xv * w0[(size_t)ci * Cout + c];
                    acc[1][c] += xv * w1[(size_t)ci * Cout + c];
                }
            }
        }
        float* o = out + ((((long long)n * Do + 2 * id + i) * Ho + 2 * ih + j) * Wo + 2 * iw) * Cout + co0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if ((CT % 4 == 0) && (Cout % 4 == 0) && co0 + CT <= Cout) {
#pragma unroll
                for (int c = 0; c < CT; c += 4) *reinterpret_cast<float4*>(o + (size_t)k * Cout + c) = make_float4(acc[k][c], acc[k][c + 1], acc[k][c + 2], acc[k][c + 3]);
            } else {
#pragma unroll
                for (int c = 0; c < CT; ++c) if (co0 + c < Cout) o[(size_t)k * Cout + c] = acc[k][c];
            }
        }
    }
}

// dx[v][ci] = sum_{tap, co} dy[2v + tap][co] * w[tap][ci][co]
template <int CT>
__global__ void __launch_bounds__(256)
deconv_k2s2_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                         int N, int D, int H, int W, int Cin, int Cout) {
    const long long nvox = (long long)N * D * H * W;
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    const int ci0 = blockIdx.y * CT;
    long long r = v;
    const int iw = (int)(r % W); r /= W;
    const int ih = (int)(r % H); r /= H;
    const int id = (int)(r % D); const int n = (int)(r / D);
    const int Ho = 2 * H, Wo = 2 * W, Do = 2 * D;
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
#pragma unroll 1
    for (int tap = 0; tap < 8; ++tap) {
        const int i = tap >> 2, j = (tap >> 1) & 1, k = tap & 1;
        const float* g = dy + ((((long long)n * Do + 2 * id + i) * Ho + 2 * ih + j) * Wo + 2 * iw + k) * Cout;
        const float* wt = w + (size_t)tap * Cin * Cout;
        for (int co = 0; co < Cout; ++co) {
            const float gv = g[co];
#pragma unroll
            for (int c = 0; c < CT; ++c) if (ci0 + c < Cin) acc[c] += gv * wt[(size_t)(ci0 + c) * Cout + co];
        }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) if (ci0 + c < Cin) dx[v * Cin + ci0 + c] = acc[c];
}

// dW[tap][ci][co] partials: each block owns a run of input voxels; thread o handles outputs o, o+256, ...
__global__ void __launch_bounds__(256)
deconv_k2s2_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ dy, float* __restrict__ partial,
                         int N, int D, int H, int W, int Cin, int Cout, long long vox_per_block) {
    const int O = 8 * Cin * Cout;
    const long long nvox = (long long)N * D * H * W;
    const long long v0 = (long long)blockIdx.x * vox_per_block;
    long long v1 = v0 + vox_per_block; if (v1 > nvox) v1 = nvox;
    const int Ho = 2 * H, Wo = 2 * W, Do = 2 * D;
    for (int o = threadIdx.x; o < O; o += blockDim.x) {
        const int co = o % Cout; const int ci = (o / Cout) % Cin; const int tap = o / (Cout * Cin);
        const int i = tap >> 2, j = (tap >> 1) & 1, k = tap & 1;
        float acc = 0.f;
        for (long long v = v0; v < v1; ++v) {
            long long r = v;
            const int iw = (int)(r % W); r /= W;
            const int ih = (int)(r % H); r /= H;
            const int id = (int)(r % D); const int n = (int)(r / D);
            acc += in[v * Cin + ci] * dy[((((long long)n * Do + 2 * id + i) * Ho + 2 * ih + j) * Wo + 2 * iw + k) * Cout + co];
        }
        partial[(size_t)blockIdx.x * O + o] = acc;
    }
}

__global__ void reduce_parts_kernel(const float* __restrict__ partial, int nparts, int O, float* __restrict__ out) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < O; o += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int b = 0; b < nparts; ++b) s += (double)partial[(size_t)b * O + o];
        out[o] = (float)s;
    }
}

// 1x1x1 conv: out[row][co] = bias[co] + sum_ci in[row][ci] * w[ci][co]   (transposed != 0: w is [Cout][Cin])
template <int CT>
__global__ void __launch_bounds__(256)
conv1x1_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
               float* __restrict__ out, long long M, int Cin, int Cout, int transposed) {
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M) return;
    const int co0 = blockIdx.y * CT;
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = (bias && co0 + c < Cout) ? bias[co0 + c] : 0.f;
    const float* x = in + row * Cin;
    for (int ci = 0; ci < Cin; ++ci) {
        const float xv = x[ci];
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (co0 + c < Cout) acc[c] += xv * (transposed ? w[(size_t)(co0 + c) * Cin + ci] : w[(size_t)ci * Cout + co0 + c]);
    }
    float* o = out + row * Cout + co0;
    if ((CT % 4 == 0) && (Cout % 4 == 0) && co0 + CT <= Cout) {
#pragma unroll
        for (int c = 0; c < CT; c += 4) *reinterpret_cast<float4*>(o + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
    } else {
#pragma unroll
        for (int c = 0; c < CT; ++c) if (co0 + c < Cout) o[c] = acc[c];
    }
}

// dW[ci][co] partials: rows staged through LDS in tiles of R rows; thread t owns outputs t, t+256, ... (<= 16 each)
constexpr int kWgR = 64;
__global__ void __launch_bounds__(256)
conv1x1_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ dy, float* __restrict__ partial,
                     long long M, int Cin, int Cout, long long rows_per_block) {
    extern __shared__ float sh[];   // [R][Cin] then [R][Cout]
    float* sx = sh; float* sg = sh + kWgR * Cin;
    const int O = Cin * Cout;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (long long base = r0; base < r1; base += kWgR) {
        const int nr = (int)((r1 - base) < kWgR ? (r1 - base) : kWgR);
        __syncthreads();
        for (int idx = threadIdx.x; idx < nr * Cin; idx += blockDim.x) sx[idx] = in[base * Cin + idx];
        for (int idx = threadIdx.x; idx < nr * Cout; idx += blockDim.x) sg[idx] = dy[base * Cout + idx];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int o = threadIdx.x + k * 256;
            if (o < O) {
                const int ci = o / Cout, co = o % Cout;
                float a = acc[k];
                for (int rr = 0; rr < nr; ++rr) a += sx[rr * Cin + ci] * sg[rr * Cout + co];
                acc[k] = a;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int o = threadIdx.x + k * 256;
        if (o < O) partial[(size_t)blockIdx.x * O + o] = acc[k];
    }
}

static int parts_for(long long units, int O, long long* per) {
    long long parts = units < 1024 ? units : 1024;
    const long long cap = (long long)(64ull << 20) / ((long long)O * 4);
    if (parts > cap) parts = cap;
    if (parts < 1) parts = 1;
    *per = da_cdiv(units, parts);
    return (int)da_cdiv(units, *per);
}

}  // namespace

extern "C" size_t da_pointwise_ws_bytes(int ntaps, int Cin, int Cout) { return da_pw_packed_bytes(ntaps, Cin, Cout) + 256; }

extern "C" int da_deconv_k2s2_fwd(const float* in, const float* w_tio, const float* bias, float* out,
                                  int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !w_tio || !out || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    const long long nvox = (long long)N * D * H * W;
    if (da_pw_supported(Cin, Cout))
        return da_pw_gemm(in, w_tio, 0, bias, out, nvox, D, H, W, Cin, Cout, 8, 1, 0, ws, ws_bytes, da_stream(stream));
    hipLaunchKernelGGL((deconv_k2s2_fwd_kernel<16>), dim3((unsigned)da_cdiv(nvox, 256), (unsigned)da_cdiv(Cout, 16)), dim3(256), 0, da_stream(stream),
                       in, w_tio, bias, out, N, D, H, W, Cin, Cout);
    DA_LAUNCH_CHECK();
    return 0;
}

// Same, also producing the BatchNorm partial sums of the output in the epilogue ([stats_nparts][2][Cout] doubles for
// da_bn_train_stats_from_partials).  *stats_nparts = 0 when the shape is not taken by the matrix-core kernel or the capacity is
// too small: the output is still computed and the caller runs da_bn_train_stats over it.
extern "C" int da_deconv_k2s2_fwd_bnstats(const float* in, const float* w_tio, const float* bias, float* out,
                                          int N, int D, int H, int W, int Cin, int Cout,
                                          double* stats_partial, int stats_capacity, int* stats_nparts,
                                          void* ws, size_t ws_bytes, void* stream) {
    if (stats_nparts) *stats_nparts = 0;
    if (!in || !w_tio || !out || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    const long long nvox = (long long)N * D * H * W;
    const long long nblk = da_cdiv(nvox, 256);
    if (stats_partial && stats_nparts && da_pw_supported(Cin, Cout) && nblk <= stats_capacity) {
        const int rc = da_pw_gemm(in, w_tio, 0, bias, out, nvox, D, H, W, Cin, Cout, 8, 1, 0, ws, ws_bytes, da_stream(stream), stats_partial);
        if (rc == 0) *stats_nparts = (int)nblk;
        return rc;
    }
    return da_deconv_k2s2_fwd(in, w_tio, bias, out, N, D, H, W, Cin, Cout, ws, ws_bytes, stream);
}

extern "C" int da_deconv_k2s2_dgrad(const float* dy, const float* w_tio, float* dx,
                                    int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !w_tio || !dx || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    const long long nvox = (long long)N * D * H * W;
    if (da_pw_supported(Cout, Cin))
        return da_pw_gemm(dy, w_tio, 1, nullptr, dx, nvox, D, H, W, Cout, Cin, 8, 1, 1, ws, ws_bytes, da_stream(stream));
    hipLaunchKernelGGL((deconv_k2s2_dgrad_kernel<16>), dim3((unsigned)da_cdiv(nvox, 256), (unsigned)da_cdiv(Cin, 16)), dim3(256), 0, da_stream(stream),
                       dy, w_tio, dx, N, D, H, W, Cin, Cout);
    DA_LAUNCH_CHECK();
    return 0;
}

// Fused backward of ConvTranspose3d(k2, s2) + BatchNorm3d(train) + LeakyReLU/ReLU (autograd of unets.py:49-52): from the gradient with respect to the
// ACTIVATED output (gout) and the raw transposed-conv output y, in one pass over the two tensors -- dx (gradient of the block's input), dW, the transposed
// conv's bias gradient, dgamma, dbeta.  The BatchNorm-backward sums come from one reduction pass over (gout, y) or from `pre` (sums the producer of gout
// accumulated: da_bn_act_bwd_dbias_pre's argument); the apply pass, the tensor dy and the data / weight gradients' second and third read of it do not exist.
// DA_ERR_UNSUPPORTED for channel counts without the fused kernel (callers then run da_bn_act_bwd_dbias + da_deconv_k2s2_dgrad + da_deconv_k2s2_wgrad).
extern "C" size_t da_deconv_k2s2_bn_bwd_ws_bytes(int N, int D, int H, int W, int Cin, int Cout) {
    const long long M = (long long)N * D * H * W;
    return da_align(da_bn_ws_bytes(M * 8, Cout)) + da_deconv_bn_bwd_ws_bytes(M, Cin, Cout);
}
extern "C" int da_deconv_k2s2_bn_bwd(const float* gout, const float* y, const float* mean, const float* rstd, const float* scale, const float* shift, float act_slope,
                                     const float* in, const float* w_tio, float* dx, float* dw_tio, float* dbias, float* dgamma, float* dbeta,
                                     int N, int D, int H, int W, int Cin, int Cout, const double* pre, int pre_n,
                                     void* ws, size_t ws_bytes, void* stream) {
    if (!gout || !y || !mean || !rstd || !scale || !shift || !in || !w_tio || !dx || !dw_tio || N <= 0 || D <= 0 || H <= 0 || W <= 0) return DA_ERR_BADARG;
    if (!da_deconv_bn_bwd_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_deconv_k2s2_bn_bwd_ws_bytes(N, D, H, W, Cin, Cout)) return DA_ERR_WS_SMALL;
    const long long M = (long long)N * D * H * W;
    hipStream_t st = da_stream(stream);
    const size_t bnb = da_align(da_bn_ws_bytes(M * 8, Cout));
    const float* cm = nullptr;
    int rc = da_bn_bwd_sums(gout, y, mean, rstd, scale, shift, act_slope, M * 8, Cout, dgamma, dbeta, &cm, ws, bnb, st, pre, pre_n);
    if (rc) return rc;
    return da_deconv_bn_bwd(gout, y, mean, rstd, scale, shift, cm, act_slope, in, w_tio, dx, dw_tio, dbias, M, D, H, W, Cin, Cout, (char*)ws + bnb, ws_bytes - bnb, st);
}

extern "C" size_t da_deconv_k2s2_wgrad_ws_bytes(int N, int D, int H, int W, int Cin, int Cout) {
    long long per;
    const int O = 8 * Cin * Cout;
    const int parts = parts_for((long long)N * D * H * W, O, &per);
    const int Cm = Cout;
    size_t direct = da_align((size_t)parts * O * sizeof(float));
    const size_t mf = da_pw_supported(Cin, Cout) ? da_pw_wgrad_ws_bytes((long long)N * D * H * W, 8, Cin, Cout) : 0;
    if (mf > direct) direct = mf;
    return direct + da_bn_ws_bytes(0, Cm) + 512;
}

extern "C" int da_deconv_k2s2_wgrad(const float* in, const float* dy, float* dw_tio, float* dbias,
                                    int N, int D, int H, int W, int Cin, int Cout,
                                    void* ws, size_t ws_bytes, void* stream) {
    if (!in || !dy || !dw_tio || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (ws_bytes < da_deconv_k2s2_wgrad_ws_bytes(N, D, H, W, Cin, Cout)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    long long per;
    const int O = 8 * Cin * Cout;
    const int parts = parts_for((long long)N * D * H * W, O, &per);
    const size_t cs_off = ((ws_bytes - da_bn_ws_bytes(0, Cout)) / 256) * 256;
    int rc_pw = DA_ERR_UNSUPPORTED;
    if (da_pw_supported(Cin, Cout)) rc_pw = da_pw_wgrad(in, dy, dw_tio, (long long)N * D * H * W, D, H, W, Cin, Cout, 8, 1, ws, cs_off, st);
    if (rc_pw != 0 && rc_pw != DA_ERR_UNSUPPORTED) return rc_pw;
    if (rc_pw == DA_ERR_UNSUPPORTED) {          // odd channel counts, or tensors beyond 32-bit byte offsets
        float* partial = (float*)ws;
        hipLaunchKernelGGL(deconv_k2s2_wgrad_kernel, dim3(parts), dim3(256), 0, st, in, dy, partial, N, D, H, W, Cin, Cout, per);
        DA_LAUNCH_CHECK();
        hipLaunchKernelGGL(reduce_parts_kernel, dim3(da_grid(O, 256)), dim3(256), 0, st, partial, parts, O, dw_tio);
        DA_LAUNCH_CHECK();
    }
    if (dbias)
        return da_colsum(dy, (long long)N * D * H * W * 8, Cout, dbias, (char*)ws + cs_off, da_bn_ws_bytes(0, Cout), stream);
    return 0;
}

extern "C" int da_conv1x1_fwd(const float* in, const float* w_io, const float* bias, float* out,
                              long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !w_io || !out || M <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (da_pw_supported(Cin, Cout))
        return da_pw_gemm(in, w_io, 0, bias, out, M, 1, 1, 1, Cin, Cout, 1, 0, 0, ws, ws_bytes, da_stream(stream));
    if (Cout >= 32) hipLaunchKernelGGL((conv1x1_kernel<32>), dim3((unsigned)da_cdiv(M, 256), (unsigned)da_cdiv(Cout, 32)), dim3(256), 0, da_stream(stream), in, w_io, bias, out, M, Cin, Cout, 0);
    else hipLaunchKernelGGL((conv1x1_kernel<8>), dim3((unsigned)da_cdiv(M, 256), (unsigned)da_cdiv(Cout, 8)), dim3(256), 0, da_stream(stream), in, w_io, bias, out, M, Cin, Cout, 0);
    DA_LAUNCH_CHECK();
    return 0;
}

// 1x1x1 convolution whose input is a RAW producer output: act(in * pro_scale + pro_shift) is what gets convolved (the deferred
// BatchNorm + LeakyReLU of the last decoder block in front of the head, unets.py:249-250).  Matrix-core path only.
extern "C" int da_conv1x1_fwd_pro(const float* in, const float* pro_scale, const float* pro_shift, float pro_slope,
                                  const float* w_io, const float* bias, float* out,
                                  long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !pro_scale || !pro_shift || !w_io || !out || M <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    return da_pw_gemm(in, w_io, 0, bias, out, M, 1, 1, 1, Cin, Cout, 1, 0, 0, ws, ws_bytes, da_stream(stream), nullptr, pro_scale, pro_shift, pro_slope);
}

extern "C" int da_conv1x1_wgrad_pro(const float* in, const float* pro_scale, const float* pro_shift, float pro_slope,
                                    const float* dy, float* dw_io, float* dbias,
                                    long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !pro_scale || !pro_shift || !dy || !dw_io || M <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv1x1_wgrad_ws_bytes(M, Cin, Cout)) return DA_ERR_WS_SMALL;
    const size_t cs_off = ((ws_bytes - da_bn_ws_bytes(0, Cout)) / 256) * 256;
    const int rc = da_pw_wgrad(in, dy, dw_io, M, 1, 1, 1, Cin, Cout, 1, 0, ws, cs_off, da_stream(stream), pro_scale, pro_shift, pro_slope);
    if (rc) return rc;
    if (dbias) return da_colsum(dy, M, Cout, dbias, (char*)ws + cs_off, da_bn_ws_bytes(0, Cout), stream);
    return 0;
}

extern "C" int da_conv1x1_dgrad(const float* dy, const float* w_io, float* dx, long long M, int Cin, int Cout,
                                void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !w_io || !dx || M <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (da_pw_supported(Cout, Cin))
        return da_pw_gemm(dy, w_io, 1, nullptr, dx, M, 1, 1, 1, Cout, Cin, 1, 0, 1, ws, ws_bytes, da_stream(stream));
    // dx[row][ci] = sum_co dy[row][co] * w[ci][co] : a 1x1 conv with "Cin" = Cout, "Cout" = Cin and w read transposed
    if (Cin >= 32) hipLaunchKernelGGL((conv1x1_kernel<32>), dim3((unsigned)da_cdiv(M, 256), (unsigned)da_cdiv(Cin, 32)), dim3(256), 0, da_stream(stream), dy, w_io, nullptr, dx, M, Cout, Cin, 1);
    else hipLaunchKernelGGL((conv1x1_kernel<8>), dim3((unsigned)da_cdiv(M, 256), (unsigned)da_cdiv(Cin, 8)), dim3(256), 0, da_stream(stream), dy, w_io, nullptr, dx, M, Cout, Cin, 1);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t da_conv1x1_wgrad_ws_bytes(long long M, int Cin, int Cout) {
    long long per;
    const int O = Cin * Cout;
    const int parts = parts_for(da_cdiv(M, kWgR), O, &per);
    size_t direct = da_align((size_t)parts * O * sizeof(float));
    const size_t mf = da_pw_supported(Cin, Cout) ? da_pw_wgrad_ws_bytes(M, 1, Cin, Cout) : 0;
    if (mf > direct) direct = mf;
    return direct + da_bn_ws_bytes(0, Cout) + 512;
}

extern "C" int da_conv1x1_wgrad(const float* in, const float* dy, float* dw_io, float* dbias,
                                long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !dy || !dw_io || M <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    const int O = Cin * Cout;
    if (O > 4096 && !da_pw_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv1x1_wgrad_ws_bytes(M, Cin, Cout)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    long long per;
    const int parts = parts_for(da_cdiv(M, kWgR), O, &per);
    const size_t cs_off = ((ws_bytes - da_bn_ws_bytes(0, Cout)) / 256) * 256;
    int rc_pw = DA_ERR_UNSUPPORTED;
    if (da_pw_supported(Cin, Cout)) rc_pw = da_pw_wgrad(in, dy, dw_io, M, 1, 1, 1, Cin, Cout, 1, 0, ws, cs_off, st);
    if (rc_pw != 0 && rc_pw != DA_ERR_UNSUPPORTED) return rc_pw;
    if (rc_pw == DA_ERR_UNSUPPORTED) {
        float* partial = (float*)ws;
        hipLaunchKernelGGL(conv1x1_wgrad_kernel, dim3(parts), dim3(256), (size_t)kWgR * (Cin + Cout) * sizeof(float), st, in, dy, partial, M, Cin, Cout, per * kWgR);
        DA_LAUNCH_CHECK();
        hipLaunchKernelGGL(reduce_parts_kernel, dim3(da_grid(O, 256)), dim3(256), 0, st, partial, parts, O, dw_io);
        DA_LAUNCH_CHECK();
    }
    if (dbias)
        return da_colsum(dy, M, Cout, dbias, (char*)ws + cs_off, da_bn_ws_bytes(0, Cout), stream);
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// nn.Conv3d(kernel 2, stride 2, padding 0) -- the generator's maxpool=False down-sampler (unets.py:231-233).  It is the adjoint
// of the k2/s2 transposed conv, so it runs on the same pointwise MFMA kernels with the roles swapped:
//   fwd  = GATHER gemm (fine x -> coarse y, + bias)   dgrad = SCATTER gemm (coarse dy -> fine dx)   wgrad = pw wgrad with
//   (in := dy coarse, dy := x fine), which yields dW as [tap][Cout][Cin].
// Weights: w_tio [8][Cin][Cout] (da_w_oik_to_tio with K3 = 8).  D, H, W are the COARSE (output) dims; the fine input is exactly
// 2D x 2H x 2W (even input sizes), channels multiples of 16.
// ---------------------------------------------------------------------------------------------------
extern "C" int da_conv_k2s2_fwd(const float* x, const float* w_tio, const float* bias, float* y,
                                int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !w_tio || !y || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    return da_pw_gemm(x, w_tio, 0, bias, y, (long long)N * D * H * W, D, H, W, Cin, Cout, 8, 1, 1, ws, ws_bytes, da_stream(stream));
}

extern "C" int da_conv_k2s2_dgrad(const float* dy, const float* w_tio, float* dx,
                                  int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !w_tio || !dx || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cout, Cin)) return DA_ERR_UNSUPPORTED;
    return da_pw_gemm(dy, w_tio, 1, nullptr, dx, (long long)N * D * H * W, D, H, W, Cout, Cin, 8, 1, 0, ws, ws_bytes, da_stream(stream));
}

extern "C" size_t da_conv_k2s2_wgrad_ws_bytes(int N, int D, int H, int W, int Cin, int Cout) {
    return da_deconv_k2s2_wgrad_ws_bytes(N, D, H, W, Cout, Cin);
}

// dw_toi: [8][Cout][Cin] (convert with da_w_tio_to_iok(dw_toi, dw_oik, Cout, Cin, 8)); dbias [Cout] optional
extern "C" int da_conv_k2s2_wgrad(const float* x, const float* dy, float* dw_toi, float* dbias,
                                  int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !dy || !dw_toi || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cout, Cin)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv_k2s2_wgrad_ws_bytes(N, D, H, W, Cin, Cout)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    const size_t cs_off = ((ws_bytes - da_bn_ws_bytes(0, Cout)) / 256) * 256;
    const int rc = da_pw_wgrad(dy, x, dw_toi, (long long)N * D * H * W, D, H, W, Cout, Cin, 8, 1, ws, cs_off, st);
    if (rc) return rc;
    if (dbias) return da_colsum(dy, (long long)N * D * H * W, Cout, dbias, (char*)ws + cs_off, da_bn_ws_bytes(0, Cout), stream);
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// bf16 activation storage (common.h).  Transposed conv k2 s2: input, output and their gradients are bf16.  1x1x1 head: its INPUT (and the
// gradient with respect to it) is bf16, the logits and their gradient stay fp32 (they are the losses' operands).  Matrix-core shapes only
// (channel counts in multiples of 16); anything else returns DA_ERR_UNSUPPORTED and the caller converts around the fp32 entry.
// ---------------------------------------------------------------------------------------------------
extern "C" int da_deconv_k2s2_fwd_bf16(const void* in, const float* w_tio, const float* bias, void* out,
                                       int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !w_tio || !out || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    return da_pw_gemm((const float*)in, w_tio, 0, bias, (float*)out, (long long)N * D * H * W, D, H, W, Cin, Cout, 8, 1, 0, ws, ws_bytes, da_stream(stream),
                      nullptr, nullptr, nullptr, -1.f, 1, 1);
}
extern "C" int da_deconv_k2s2_fwd_bnstats_bf16(const void* in, const float* w_tio, const float* bias, void* out,
                                               int N, int D, int H, int W, int Cin, int Cout,
                                               double* stats_partial, int stats_capacity, int* stats_nparts,
                                               void* ws, size_t ws_bytes, void* stream) {
    if (stats_nparts) *stats_nparts = 0;
    if (!in || !w_tio || !out || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    const long long nvox = (long long)N * D * H * W, nblk = da_cdiv(nvox, 256);
    const bool stats = stats_partial && stats_nparts && nblk <= stats_capacity;
    const int rc = da_pw_gemm((const float*)in, w_tio, 0, bias, (float*)out, nvox, D, H, W, Cin, Cout, 8, 1, 0, ws, ws_bytes, da_stream(stream),
                              stats ? stats_partial : nullptr, nullptr, nullptr, -1.f, 1, 1);
    if (rc == 0 && stats) *stats_nparts = (int)nblk;
    return rc;
}
extern "C" int da_deconv_k2s2_dgrad_bf16(const void* dy, const float* w_tio, void* dx,
                                         int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !w_tio || !dx || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cout, Cin)) return DA_ERR_UNSUPPORTED;
    return da_pw_gemm((const float*)dy, w_tio, 1, nullptr, (float*)dx, (long long)N * D * H * W, D, H, W, Cout, Cin, 8, 1, 1, ws, ws_bytes, da_stream(stream),
                      nullptr, nullptr, nullptr, -1.f, 1, 1);
}
extern "C" int da_deconv_k2s2_wgrad_bf16(const void* in, const void* dy, float* dw_tio, float* dbias,
                                         int N, int D, int H, int W, int Cin, int Cout,
                                         void* ws, size_t ws_bytes, void* stream) {
    if (!in || !dy || !dw_tio || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_deconv_k2s2_wgrad_ws_bytes(N, D, H, W, Cin, Cout)) return DA_ERR_WS_SMALL;
    const size_t cs_off = ((ws_bytes - da_bn_ws_bytes(0, Cout)) / 256) * 256;
    const int rc = da_pw_wgrad((const float*)in, (const float*)dy, dw_tio, (long long)N * D * H * W, D, H, W, Cin, Cout, 8, 1, ws, cs_off, da_stream(stream),
                               nullptr, nullptr, -1.f, 1, 1);
    if (rc) return rc;
    if (dbias) return da_colsum_bf16(dy, (long long)N * D * H * W * 8, Cout, dbias, (char*)ws + cs_off, da_bn_ws_bytes(0, Cout), stream);
    return 0;
}

extern "C" int da_conv1x1_fwd_bf16(const void* in, const float* w_io, const float* bias, float* out,
                                   long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !w_io || !out || M <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    return da_pw_gemm((const float*)in, w_io, 0, bias, out, M, 1, 1, 1, Cin, Cout, 1, 0, 0, ws, ws_bytes, da_stream(stream), nullptr, nullptr, nullptr, -1.f, 1, 0);
}
extern "C" int da_conv1x1_fwd_pro_bf16(const void* in, const float* pro_scale, const float* pro_shift, float pro_slope,
                                       const float* w_io, const float* bias, float* out,
                                       long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !pro_scale || !pro_shift || !w_io || !out || M <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    return da_pw_gemm((const float*)in, w_io, 0, bias, out, M, 1, 1, 1, Cin, Cout, 1, 0, 0, ws, ws_bytes, da_stream(stream), nullptr, pro_scale, pro_shift, pro_slope, 1, 0);
}
extern "C" int da_conv1x1_dgrad_bf16(const float* dy, const float* w_io, void* dx, long long M, int Cin, int Cout,
                                     void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !w_io || !dx || M <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cout, Cin)) return DA_ERR_UNSUPPORTED;
    return da_pw_gemm(dy, w_io, 1, nullptr, (float*)dx, M, 1, 1, 1, Cout, Cin, 1, 0, 1, ws, ws_bytes, da_stream(stream), nullptr, nullptr, nullptr, -1.f, 0, 1);
}
extern "C" int da_conv1x1_wgrad_bf16(const void* in, const float* dy, float* dw_io, float* dbias,
                                     long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !dy || !dw_io || M <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv1x1_wgrad_ws_bytes(M, Cin, Cout)) return DA_ERR_WS_SMALL;
    const size_t cs_off = ((ws_bytes - da_bn_ws_bytes(0, Cout)) / 256) * 256;
    const int rc = da_pw_wgrad((const float*)in, dy, dw_io, M, 1, 1, 1, Cin, Cout, 1, 0, ws, cs_off, da_stream(stream), nullptr, nullptr, -1.f, 1, 0);
    if (rc) return rc;
    if (dbias) return da_colsum(dy, M, Cout, dbias, (char*)ws + cs_off, da_bn_ws_bytes(0, Cout), stream);
    return 0;
}
extern "C" int da_conv1x1_wgrad_pro_bf16(const void* in, const float* pro_scale, const float* pro_shift, float pro_slope,
                                         const float* dy, float* dw_io, float* dbias,
                                         long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !pro_scale || !pro_shift || !dy || !dw_io || M <= 0 || Cin <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (!da_pw_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv1x1_wgrad_ws_bytes(M, Cin, Cout)) return DA_ERR_WS_SMALL;
    const size_t cs_off = ((ws_bytes - da_bn_ws_bytes(0, Cout)) / 256) * 256;
    const int rc = da_pw_wgrad((const float*)in, dy, dw_io, M, 1, 1, 1, Cin, Cout, 1, 0, ws, cs_off, da_stream(stream), pro_scale, pro_shift, pro_slope, 1, 0);
    if (rc) return rc;
    if (dbias) return da_colsum(dy, M, Cout, dbias, (char*)ws + cs_off, da_bn_ws_bytes(0, Cout), stream);
    return 0;
}
