// 3x3x3 convolution (padding 1, stride 1|2): C-ABI dispatch + the direct (VALU) kernels.
// Rows a1/a7/a9 of SURVEY.md §8: nn.Conv3d(k=3,p=1) at unets.py:30,36, modules.py:48, voxel_morph.py:57.
//
// Two families live behind da_conv3d_k3_{fwd,dgrad,wgrad}:
//   * conv3d_mfma.hip : implicit-GEMM on v_mfma_f32_16x16x4_f32 (exact fp32) with LDS-staged halo tiles --
//                       the MFMA-bound path for Cin % 8 == 0 layers, which carry >95 % of the FLOPs.
//   * this file       : direct kernels -- one output voxel per lane, all (or a group of) output channels in
//                       registers, weights fetched through wave-uniform (scalar) loads.  Used where the layer
//                       is HBM-bound or tiny (Cin in {1,2,3}, Cout = 3, stride-2 dgrad) and as the reference
//                       implementation the MFMA kernels are A/B-checked against on the GPU.
#include "common.h"
#include "conv3d_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// forward, direct.  in = concat(in1[C1], in2[C2]); out channels [0,Cs1) -> out1, [Cs1, Cout) -> out2.
// ---------------------------------------------------------------------------------------------------
template <int CT>
__global__ void __launch_bounds__(256)
conv3_direct_fwd_kernel(const float* __restrict__ in1, int C1, const float* __restrict__ in2, int C2,
                        const float* __restrict__ w, const float* __restrict__ bias,
                        float* __restrict__ out1, int Cs1, float* __restrict__ out2, int Cs2,
                        int N, int D, int H, int W, int Do, int Ho, int Wo, int Cout, int stride, float slope) {
    const long long nvox = (long long)N * Do * Ho * Wo;
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    const int co0 = blockIdx.y * CT;
    const int Cin = C1 + C2;
    long long r = v;
    const int ow = (int)(r % Wo); r /= Wo;
    const int oh = (int)(r % Ho); r /= Ho;
    const int od = (int)(r % Do); const int n = (int)(r / Do);
    float acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = (bias && co0 + j < Cout) ? bias[co0 + j] : 0.f;
    for (int kd = 0; kd < 3; ++kd) {
        const int id = od * stride - 1 + kd;
        if (id < 0 || id >= D) continue;
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * stride - 1 + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * stride - 1 + kw;
                if (iw < 0 || iw >= W) continue;
                const int tap = (kd * 3 + kh) * 3 + kw;
                const long long iv = (((long long)n * D + id) * H + ih) * W + iw;
                const float* wt = w + (size_t)tap * Cin * Cout + co0;
                {
                    const float* p = in1 + iv * C1;
                    if ((C1 & 3) == 0) {
                        for (int ci = 0; ci < C1; ci += 4) {
                            const float4 x = *reinterpret_cast<const float4*>(p + ci);
#pragma unroll
                            for (int j = 0; j < CT; ++j) {
                                if (co0 + j < Cout) {
                                    acc[j] += x.x * wt[(size_t)(ci + 0) * Cout + j];
                                    acc[j] += x.y * wt[(size_t)(ci + 1) * Cout + j];
                                    acc[j] += x.z * wt[(size_t)(ci + 2) * Cout + j];
                                    acc[j] += x.w * wt[(size_t)(ci + 3) * Cout + j];
                                }
                            }
                        }
                    } else {
                        for (int ci = 0; ci < C1; ++ci) {
                            const float x = p[ci];
#pragma unroll
                            for (int j = 0; j < CT; ++j) if (co0 + j < Cout) acc[j] += x * wt[(size_t)ci * Cout + j];
                        }
                    }
                }
                if (C2 > 0) {
                    const float* p = in2 + iv * C2;
                    const float* wt2 = wt + (size_t)C1 * Cout;
                    if ((C2 & 3) == 0) {
                        for (int ci = 0; ci < C2; ci += 4) {
                            const float4 x = *reinterpret_cast<const float4*>(p + ci);
#pragma unroll
                            for (int j = 0; j < CT; ++j) {
                                if (co0 + j < Cout) {
                                    acc[j] += x.x * wt2[(size_t)(ci + 0) * Cout + j];
                                    acc[j] += x.y * wt2[(size_t)(ci + 1) * Cout + j];
                                    acc[j] += x.z * wt2[(size_t)(ci + 2) * Cout + j];
                                    acc[j] += x.w * wt2[(size_t)(ci + 3) * Cout + j];
                                }
                            }
                        }
                    } else {
                        for (int ci = 0; ci < C2; ++ci) {
                            const float x = p[ci];
#pragma unroll
                            for (int j = 0; j < CT; ++j) if (co0 + j < Cout) acc[j] += x * wt2[(size_t)ci * Cout + j];
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = da_act(acc[j], slope);
    // stores: 16-byte when the whole group sits in one destination and is aligned
    const bool vec = (CT % 4 == 0) && (co0 + CT <= Cout) && ((co0 + CT <= Cs1 && (Cs1 & 3) == 0) || (co0 >= Cs1 && (Cs2 & 3) == 0 && ((co0 - Cs1) & 3) == 0));
    if (vec) {
        float* o = (co0 + CT <= Cs1) ? out1 + v * Cs1 + co0 : out2 + v * Cs2 + (co0 - Cs1);
#pragma unroll
        for (int j = 0; j + 3 < CT; j += 4) *reinterpret_cast<float4*>(o + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int c = co0 + j;
            if (c < Cout) { if (c < Cs1) out1[v * Cs1 + c] = acc[j]; else out2[v * Cs2 + (c - Cs1)] = acc[j]; }
        }
    }
}

// Forward for Cin <= 2 (seg enc0.0: 1 -> 8, reg enc0: (source, target) -> 16): HBM-bound, 27*Cin input taps per voxel.
// All taps are fetched through buffer descriptors (out-of-volume tap -> offset 0xFFFFFFFF -> 0) before any FMA, so the
// loads issue back to back instead of one exec-mask branch + wait per tap; weights come through scalar loads.
template <int CT>
__global__ void __launch_bounds__(256)
conv3_tinycin_fwd_kernel(const float* __restrict__ in1, int C1, const float* __restrict__ in2, int C2,
                         const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out,
                         int N, int D, int H, int W, int Cout, float slope) {
    const long long nvox = (long long)N * D * H * W;
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = v < nvox;
    const long long vv = live ? v : 0;
    long long r = vv;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H); r /= H;
    const int z = (int)(r % D); const int n = (int)(r / D);
    const int Cin = C1 + C2;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)in1, 0, (unsigned)((unsigned long long)nvox * C1 * 4ull), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(C2 > 0 ? in2 : in1), 0, (unsigned)((unsigned long long)nvox * (C2 > 0 ? C2 : C1) * 4ull), 0x00020000);
    float xin[27][2];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const int zz = z + tap / 9 - 1, yy = y + (tap / 3) % 3 - 1, xx = x + tap % 3 - 1;
        const bool inb = live && (unsigned)zz < (unsigned)D && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        const unsigned vox = (unsigned)(((n * D + zz) * H + yy) * W + xx);
        if (C2 > 0) {          // one channel from each pointer
            xin[tap][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, inb ? vox * 4u : 0xFFFFFFFFu, 0, 0));
            xin[tap][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, inb ? vox * 4u : 0xFFFFFFFFu, 0, 0));
        } else if (C1 == 2) {
            xin[tap][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, inb ? vox * 8u : 0xFFFFFFFFu, 0, 0));
            xin[tap][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, inb ? vox * 8u + 4u : 0xFFFFFFFFu, 0, 0));
        } else {
            xin[tap][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, inb ? vox * 4u : 0xFFFFFFFFu, 0, 0));
            xin[tap][1] = 0.f;
        }
    }
    float acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = (bias && j < Cout) ? bias[j] : 0.f;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const float* wt = w + (size_t)tap * Cin * Cout;
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            if (j < Cout) {
                acc[j] += xin[tap][0] * wt[j];
                if (Cin == 2) acc[j] += xin[tap][1] * wt[Cout + j];
            }
        }
    }
    if (!live) return;
    float* o = out + v * Cout;
#pragma unroll
    for (int j = 0; j < CT; j += 4)
        if (j + 3 < Cout) *reinterpret_cast<float4*>(o + j) = make_float4(da_act(acc[j], slope), da_act(acc[j + 1], slope), da_act(acc[j + 2], slope), da_act(acc[j + 3], slope));
}

// w_tio [27][Cin][Cout] -> flipped + transposed [27][Cout][Cin]: stride-1 dgrad is a forward conv with it.
__global__ void w_flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wf, int Cin, int Cout) {
    const int total = 27 * Cin * Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % Cout; const int ci = (i / Cout) % Cin; const int t = i / (Cout * Cin);
        wf[((size_t)(26 - t) * Cout + co) * Cin + ci] = w[i];
    }
}

// stride-2 data gradient (gather): dx[p][ci] = sum_{t, co : 2o - 1 + t = p} dy[o][co] * w[t][ci][co]
template <int CT>
__global__ void __launch_bounds__(256)
conv3_s2_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                      float* __restrict__ dx1, int C1, float* __restrict__ dx2, int C2,
                      int N, int D, int H, int W, int Do, int Ho, int Wo, int Cout) {
    const long long nvox = (long long)N * D * H * W;
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    const int Cin = C1 + C2;
    const int ci0 = blockIdx.y * CT;
    long long r = v;
    const int iw = (int)(r % W); r /= W;
    const int ih = (int)(r % H); r /= H;
    const int id = (int)(r % D); const int n = (int)(r / D);
    float acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = 0.f;
    for (int kd = 0; kd < 3; ++kd) {
        const int td = id + 1 - kd;
        if (td < 0 || (td & 1)) continue;
        const int od = td >> 1; if (od >= Do) continue;
        for (int kh = 0; kh < 3; ++kh) {
            const int th = ih + 1 - kh;
            if (th < 0 || (th & 1)) continue;
            const int oh = th >> 1; if (oh >= Ho) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int tw = iw + 1 - kw;
                if (tw < 0 || (tw & 1)) continue;
                const int ow = tw >> 1; if (ow >= Wo) continue;
                const int tap = (kd * 3 + kh) * 3 + kw;
                const float* g = dy + ((((long long)n * Do + od) * Ho + oh) * Wo + ow) * Cout;
                const float* wt = w + (size_t)tap * Cin * Cout;
                for (int co = 0; co < Cout; ++co) {
                    const float gv = g[co];
#pragma unroll
                    for (int j = 0; j < CT; ++j) if (ci0 + j < Cin) acc[j] += gv * wt[(size_t)(ci0 + j) * Cout + co];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int c = ci0 + j;
        if (c < Cin) { if (c < C1) dx1[v * C1 + c] = acc[j]; else dx2[v * C2 + (c - C1)] = acc[j]; }
    }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient, direct: every block owns a run of output rows (n, od, oh) and produces a partial
// dW[27][Cin][Cout]; a second launch sums the partials in double (deterministic).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
conv3_direct_wgrad_kernel(const float* __restrict__ in1, int C1, const float* __restrict__ in2, int C2,
                          const float* __restrict__ dy, float* __restrict__ partial,
                          int N, int D, int H, int W, int Do, int Ho, int Wo, int Cout, int stride, int rows_per_block) {
    const int Cin = C1 + C2;
    const int O = 27 * Cin * Cout;
    const long long nrows = (long long)N * Do * Ho;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block; if (r1 > nrows) r1 = nrows;
    for (int o = threadIdx.x; o < O; o += blockDim.x) {
        const int co = o % Cout; const int ci = (o / Cout) % Cin; const int tap = o / (Cout * Cin);
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        const float* src; int Cs, cs;
        if (ci < C1) { src = in1; Cs = C1; cs = ci; } else { src = in2; Cs = C2; cs = ci - C1; }
        float acc = 0.f;
        for (long long row = r0; row < r1; ++row) {
            const int oh = (int)(row % Ho); const int od = (int)((row / Ho) % Do); const int n = (int)(row / ((long long)Ho * Do));
            const int id = od * stride - 1 + kd, ih = oh * stride - 1 + kh;
            if (id < 0 || id >= D || ih < 0 || ih >= H) continue;
            const float* g = dy + (row * Wo) * Cout + co;
            const float* x = src + ((((long long)n * D + id) * H + ih) * W) * Cs + cs;
            for (int ow = 0; ow < Wo; ++ow) {
                const int iw = ow * stride - 1 + kw;
                if (iw < 0 || iw >= W) continue;
                acc += g[(long long)ow * Cout] * x[(long long)iw * Cs];
            }
        }
        partial[(size_t)blockIdx.x * O + o] = acc;
    }
}

__global__ void partial_reduce_kernel(const float* __restrict__ partial, int nparts, int O, float* __restrict__ out) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < O; o += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int b = 0; b < nparts; ++b) s += (double)partial[(size_t)b * O + o];
        out[o] = (float)s;
    }
}

template <int CT, typename... Args>
static int launch_fwd(long long nvox, int Cout, hipStream_t st, Args... args) {
    hipLaunchKernelGGL((conv3_direct_fwd_kernel<CT>), dim3((unsigned)da_cdiv(nvox, 256), (unsigned)da_cdiv(Cout, CT)), dim3(256), 0, st, args...);
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// ---- kept packed operands of the families outside the split matrix kernels (conv3d_internal.h: da_pp_lookup) --------------------------------------
namespace {
struct KeptAny { int mode; const float* w; unsigned char* buf; size_t cap; int tag; size_t need; int filled; };      // mode 0 none, 1 hand-over (use), 2 fill
thread_local KeptAny g_kp = {0, nullptr, nullptr, 0, 0, 0, 0};
}
DaKeptPack da_pp_lookup(const float* w_tio, size_t need, int tag) {
    if (g_kp.mode == 2) {                                       // da_conv3d_k3_prepack_any in progress: size query (no buffer) or fill
        g_kp.need = need; g_kp.tag = tag;
        if (g_kp.w == w_tio && g_kp.buf && g_kp.cap >= need) { g_kp.filled = 1; return DaKeptPack{g_kp.buf, 1, 1}; }
        return DaKeptPack{nullptr, 0, 1};
    }
    if (g_kp.mode == 1) {
        const bool ok = g_kp.w == w_tio && g_kp.tag == tag && g_kp.buf && g_kp.cap >= need;
        unsigned char* b = ok ? g_kp.buf : nullptr;
        g_kp.mode = 0;
        return DaKeptPack{b, 0, 0};
    }
    return DaKeptPack{nullptr, 0, 0};
}
void da_pp_drop_handover() { if (g_kp.mode == 1) g_kp.mode = 0; }

// ---------------------------------------------------------------------------------------------------
// internal entry points (also used by the MFMA dispatcher and the tests' A/B switch)
// ---------------------------------------------------------------------------------------------------
int da_conv3_direct_fwd(const float* in1, int C1, const float* in2, int C2, const float* w, const float* bias,
                        float* out1, int Cs1, float* out2, int Cs2,
                        int N, int D, int H, int W, int Cout, int stride, float slope, hipStream_t st) {
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long long nvox = (long long)N * Do * Ho * Wo;
    if (Cout <= 4)  return launch_fwd<4>(nvox, Cout, st, in1, C1, in2, C2, w, bias, out1, Cs1, out2, Cs2, N, D, H, W, Do, Ho, Wo, Cout, stride, slope);
    if (Cout <= 8)  return launch_fwd<8>(nvox, Cout, st, in1, C1, in2, C2, w, bias, out1, Cs1, out2, Cs2, N, D, H, W, Do, Ho, Wo, Cout, stride, slope);
    return launch_fwd<16>(nvox, Cout, st, in1, C1, in2, C2, w, bias, out1, Cs1, out2, Cs2, N, D, H, W, Do, Ho, Wo, Cout, stride, slope);
}

static size_t wgrad_direct_parts(long long nrows, int O, int* rows_per_block) {
    long long parts = nrows < 1024 ? nrows : 1024;
    const long long cap = (long long)(64ull << 20) / ((long long)O * 4);
    if (parts > cap) parts = cap;
    if (parts < 1) parts = 1;
    *rows_per_block = (int)da_cdiv(nrows, parts);
    return (size_t)da_cdiv(nrows, *rows_per_block);
}

extern "C" size_t da_conv3d_k3_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int stride) {
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1;
    const int O = 27 * Cin * Cout;
    int rpb;
    const size_t parts = wgrad_direct_parts((long long)N * Do * Ho, O, &rpb);
    size_t direct = da_align(parts * (size_t)O * sizeof(float));
    size_t mfma = da_conv3_mfma_ws_bytes(N, D, H, W, Cin, Cout, stride);
    if (stride == 2 && da_conv3_s2_supported(Cin, 0, Cout)) mfma = da_conv3_s2_ws_bytes(N, D, H, W, Cin, Cout);
    size_t packed = da_align((size_t)2 * O * sizeof(float) + 65536);
    const int Cm = Cin > Cout ? Cin : Cout;
    return packed + (direct > mfma ? direct : mfma) + da_bn_ws_bytes(0, Cm) + 4096;
}

static int g_force_direct = -1;
static bool force_direct() {
    if (g_force_direct < 0) { const char* e = getenv("DA_CONV_DIRECT"); g_force_direct = (e && e[0] == '1') ? 1 : 0; }
    return g_force_direct == 1;
}
extern "C" int da_set_conv_direct(int on) { const int prev = force_direct() ? 1 : 0; g_force_direct = on ? 1 : 0; return prev; }

// Coarse stride-2 layers (the registration encoder below 40^3) are latency-bound on the space-to-depth route (a chain of four launches,
// 16 channel chunks per tile walked one after the other: 0.15 - 0.28 ms for 0.05 - 0.4 GFLOP).  Measured alternative: the direct
// kernels are 2 - 8x SLOWER there (forward 0.33 vs 0.15 ms, data gradient 0.52 vs 0.05, weight gradient 0.58 vs 0.07 at 40^3 -> 20^3),
// so the switch stays off: DA_S2_DIRECT_MAX_OUT_VOX = largest output-voxel count that takes the direct kernels (default 0 = never).
static bool s2_prefers_direct(int N, int D, int H, int W) {
    static long long thr = -1;
    if (thr < 0) { const char* e = getenv("DA_S2_DIRECT_MAX_OUT_VOX"); thr = e ? atoll(e) : 0; }
    const long long out = (long long)N * ((D - 1) / 2 + 1) * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
    return out <= thr;
}

extern "C" int da_conv3d_k3_fwd(const float* in1, int C1, const float* in2, int C2,
                                const float* w_tio, const float* bias, float* out,
                                int N, int D, int H, int W, int Cout, int stride, float act_slope,
                                void* ws, size_t ws_bytes, void* stream) {
    if (!in1 || !w_tio || !out || C1 <= 0 || C2 < 0 || (C2 > 0 && !in2) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (stride != 1 && stride != 2))
        return DA_ERR_BADARG;
    hipStream_t st = da_stream(stream);
    DaPpScope pp_scope;
    if (!force_direct() && stride == 2 && da_conv3_s2_supported(C1, C2, Cout) && !s2_prefers_direct(N, D, H, W)) {
        if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, stride)) return DA_ERR_WS_SMALL;
        return da_conv3_s2_fwd(in1, C1, w_tio, bias, out, N, D, H, W, Cout, act_slope, ws, ws_bytes - da_bn_ws_bytes(0, Cout) - 4096, st);
    }
    if (!force_direct() && da_conv3_mfma_fwd_supported(C1, C2, Cout, stride)) {
        if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, stride)) return DA_ERR_WS_SMALL;
        const int rc = da_conv3_mfma_fwd(in1, C1, in2, C2, w_tio, /*w_is_flipped_tr=*/0, bias, out, Cout, nullptr, 0,
                                         N, D, H, W, Cout, stride, act_slope, ws, ws_bytes, st);
        if (rc != DA_ERR_UNSUPPORTED) return rc;           // e.g. a per-sample tensor beyond 32-bit byte offsets: direct kernels below
    }
    if (!force_direct() && stride == 1 && da_conv3_flowmm_supported(C1, C2, Cout, N, D, H, W)) {      // <= 3 outputs, split mode: (dy, cout) columns on the matrix cores
        const int rc = da_conv3_flowmm_fwd(in1, C1, in2, C2, w_tio, bias, out, N, D, H, W, Cout, act_slope, ws, ws_bytes, st);
        if (rc != DA_ERR_UNSUPPORTED) return rc;
    }
    if (!force_direct() && da_conv3_thin_supported(C1, C2, Cout, stride) && !getenv("DA_NO_THIN")) {
        const int rc = da_conv3_thin_fwd(in1, C1, in2, C2, w_tio, 0, bias, out, Cout, nullptr, 0, N, D, H, W, Cout, act_slope, ws, ws_bytes, st);
        if (rc != DA_ERR_UNSUPPORTED) return rc;
    }
    // tiny Cin, full-quad Cout, single output pointer, 32-bit addressable
    const bool tiny = !force_direct() && stride == 1 && ((C2 == 0 && (C1 == 1 || C1 == 2)) || (C1 == 1 && C2 == 1)) && (Cout == 8 || Cout == 16) &&
                      (unsigned long long)N * D * H * W * 2ull * 4ull < 0xFFFFFFF0ull;
    if (tiny) {
        const long long nvox = (long long)N * D * H * W;
        if (Cout == 8) hipLaunchKernelGGL((conv3_tinycin_fwd_kernel<8>), dim3((unsigned)da_cdiv(nvox, 256)), dim3(256), 0, st, in1, C1, in2, C2, w_tio, bias, out, N, D, H, W, Cout, act_slope);
        else hipLaunchKernelGGL((conv3_tinycin_fwd_kernel<16>), dim3((unsigned)da_cdiv(nvox, 256)), dim3(256), 0, st, in1, C1, in2, C2, w_tio, bias, out, N, D, H, W, Cout, act_slope);
        DA_LAUNCH_CHECK();
        return 0;
    }
    return da_conv3_direct_fwd(in1, C1, in2, C2, w_tio, bias, out, Cout, nullptr, 0, N, D, H, W, Cout, stride, act_slope, st);
}

extern "C" int da_conv3d_k3_fwd_bnstats(const float* in1, int C1, const float* in2, int C2,
                                        const float* w_tio, const float* bias, float* out,
                                        int N, int D, int H, int W, int Cout, int stride,
                                        double* stats_partial, int stats_capacity, int* stats_nparts,
                                        void* ws, size_t ws_bytes, void* stream) {
    if (stats_nparts) *stats_nparts = 0;
    if (!in1 || !w_tio || !out || C1 <= 0 || C2 < 0 || (C2 > 0 && !in2) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (stride != 1 && stride != 2))
        return DA_ERR_BADARG;
    if (stats_partial && stats_capacity >= 512 && !force_direct() && stride == 1 && da_conv3_mfma_fwd_supported(C1, C2, Cout, stride)) {
        if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, stride)) return DA_ERR_WS_SMALL;
        const int rc = da_conv3_mfma_fwd(in1, C1, in2, C2, w_tio, 0, bias, out, Cout, nullptr, 0, N, D, H, W, Cout, stride, -1.f,
                                         ws, ws_bytes, da_stream(stream), 0, stats_partial, stats_nparts);
        if (rc != DA_ERR_UNSUPPORTED) return rc;
        if (stats_nparts) *stats_nparts = 0;
    }
    return da_conv3d_k3_fwd(in1, C1, in2, C2, w_tio, bias, out, N, D, H, W, Cout, stride, -1.f, ws, ws_bytes, stream);
}

// Forward / weight gradient with an INPUT PROLOGUE: in1 / in2 may be raw outputs of a BatchNorm'd producer whose per-channel
// scale / shift and activation are applied while the tile is staged (the activated tensor is never written to HBM).  Matrix-core
// path only, stride 1; DA_ERR_UNSUPPORTED = the caller materialises the activation (da_bn_act_fwd) and uses the plain entry.
extern "C" int da_conv3d_k3_fwd_pro(const float* in1, int C1, const float* pro1_scale, const float* pro1_shift, float pro1_slope,
                                    const float* in2, int C2, const float* pro2_scale, const float* pro2_shift, float pro2_slope,
                                    const float* w_tio, const float* bias, float* out,
                                    int N, int D, int H, int W, int Cout, float act_slope,
                                    double* stats_partial, int stats_capacity, int* stats_nparts,
                                    void* ws, size_t ws_bytes, void* stream) {
    if (stats_nparts) *stats_nparts = 0;
    if (!in1 || !w_tio || !out || C1 <= 0 || C2 < 0 || (C2 > 0 && !in2) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 ||
        (pro1_scale && !pro1_shift) || (pro2_scale && !pro2_shift))
        return DA_ERR_BADARG;
    if (force_direct() || !da_conv3_mfma_fwd_supported(C1, C2, Cout, 1)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, 1)) return DA_ERR_WS_SMALL;
    const DaPro pro = {pro1_scale, pro1_shift, pro1_slope, pro2_scale, pro2_shift, pro2_slope};
    const bool stats = stats_partial && stats_capacity >= 512;
    return da_conv3_mfma_fwd(in1, C1, in2, C2, w_tio, 0, bias, out, Cout, nullptr, 0, N, D, H, W, Cout, 1, stats ? -1.f : act_slope,
                             ws, ws_bytes, da_stream(stream), 0, stats ? stats_partial : nullptr, stats ? stats_nparts : nullptr, &pro);
}

extern "C" int da_conv3d_k3_wgrad_pro(const float* in1, int C1, const float* pro1_scale, const float* pro1_shift, float pro1_slope,
                                      const float* in2, int C2, const float* pro2_scale, const float* pro2_shift, float pro2_slope,
                                      const float* dy, float* dw_tio,
                                      int N, int D, int H, int W, int Cout,
                                      void* ws, size_t ws_bytes, void* stream) {
    if (!in1 || !dy || !dw_tio || C1 <= 0 || C2 < 0 || (C2 > 0 && !in2) || N <= 0 || Cout <= 0 ||
        (pro1_scale && !pro1_shift) || (pro2_scale && !pro2_shift))
        return DA_ERR_BADARG;
    if (force_direct() || !da_conv3_mfma_wgrad_supported(C1, C2, Cout, 1)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, 1)) return DA_ERR_WS_SMALL;
    const DaPro pro = {pro1_scale, pro1_shift, pro1_slope, pro2_scale, pro2_shift, pro2_slope};
    return da_conv3_mfma_wgrad(in1, C1, in2, C2, dy, dw_tio, N, D, H, W, Cout, 1, ws, ws_bytes, da_stream(stream), 0, &pro);
}

extern "C" int da_conv3d_k3_dgrad(const float* dy, const float* w_tio, float* dx1, int C1, float* dx2, int C2,
                                  int N, int D, int H, int W, int Cout, int stride,
                                  void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !w_tio || !dx1 || C1 <= 0 || C2 < 0 || (C2 > 0 && !dx2) || N <= 0 || Cout <= 0 || (stride != 1 && stride != 2)) return DA_ERR_BADARG;
    hipStream_t st = da_stream(stream);
    DaPpScope pp_scope;
    const int Cin = C1 + C2;
    if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, Cin, Cout, stride)) return DA_ERR_WS_SMALL;
    if (stride == 1) {
        // dX = conv(dY, flip/transpose(W)) : Cin' = Cout, Cout' = Cin, output split over (dx1, dx2)
        if (!force_direct() && da_conv3_mfma_fwd_supported(Cout, 0, Cin, 1, C1, C2)) {
            const int rc = da_conv3_mfma_fwd(dy, Cout, nullptr, 0, w_tio, /*w_is_flipped_tr=*/1, nullptr, dx1, C1, dx2, C2,
                                             N, D, H, W, Cin, 1, -1.f, ws, ws_bytes, st);
            if (rc != DA_ERR_UNSUPPORTED) return rc;
        }
        if (!force_direct() && da_conv3_flowmm_supported(C1, C2, Cout, N, D, H, W)) {
            const int rc = da_conv3_flowmm_dgrad(dy, w_tio, dx1, C1, dx2, C2, N, D, H, W, Cout, ws, ws_bytes, st);
            if (rc != DA_ERR_UNSUPPORTED) return rc;
        }
        if (!force_direct() && da_conv3_thin_supported(Cout, 0, Cin, 1)) {
            const int rc = da_conv3_thin_fwd(dy, Cout, nullptr, 0, w_tio, 1, nullptr, dx1, C1, dx2, C2, N, D, H, W, Cin, -1.f, ws, ws_bytes, st);
            if (rc != DA_ERR_UNSUPPORTED) return rc;
        }
        float* wf = (float*)ws;
        hipLaunchKernelGGL(w_flip_transpose_kernel, dim3(da_grid(27 * Cin * Cout, 256)), dim3(256), 0, st, w_tio, wf, Cin, Cout);
        DA_LAUNCH_CHECK();
        return da_conv3_direct_fwd(dy, Cout, nullptr, 0, wf, nullptr, dx1, C1, dx2, C2, N, D, H, W, Cin, 1, -1.f, st);
    }
    if (!force_direct() && da_conv3_s2_supported(C1, C2, Cout) && !s2_prefers_direct(N, D, H, W))
        return da_conv3_s2_dgrad(dy, w_tio, dx1, C1, N, D, H, W, Cout, ws, ws_bytes - da_bn_ws_bytes(0, Cout) - 4096, st);
    const int Do = (D - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long nvox = (long long)N * D * H * W;
    hipLaunchKernelGGL((conv3_s2_dgrad_kernel<16>), dim3((unsigned)da_cdiv(nvox, 256), (unsigned)da_cdiv(Cin, 16)), dim3(256), 0, st,
                       dy, w_tio, dx1, C1, dx2, C2, N, D, H, W, Do, Ho, Wo, Cout);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_conv3d_k3_wgrad(const float* in1, int C1, const float* in2, int C2, const float* dy,
                                  float* dw_tio, float* dbias,
                                  int N, int D, int H, int W, int Cout, int stride,
                                  void* ws, size_t ws_bytes, void* stream) {
    if (!in1 || !dy || !dw_tio || C1 <= 0 || C2 < 0 || (C2 > 0 && !in2) || N <= 0 || Cout <= 0 || (stride != 1 && stride != 2)) return DA_ERR_BADARG;
    hipStream_t st = da_stream(stream);
    const int Cin = C1 + C2;
    if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, Cin, Cout, stride)) return DA_ERR_WS_SMALL;
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int O = 27 * Cin * Cout;
    int rc = 0;
    if (!force_direct() && stride == 2 && da_conv3_s2_supported(C1, C2, Cout) && !s2_prefers_direct(N, D, H, W)) {
        rc = da_conv3_s2_wgrad(in1, C1, dy, dw_tio, N, D, H, W, Cout, ws, ws_bytes - da_bn_ws_bytes(0, Cout) - 4096, st);
    } else if (!force_direct() && stride == 1 && da_conv3_mfma_wgrad_supported(C1, C2, Cout, stride)) {
        rc = da_conv3_mfma_wgrad(in1, C1, in2, C2, dy, dw_tio, N, D, H, W, Cout, stride, ws, ws_bytes, st);
    } else {
        rc = DA_ERR_UNSUPPORTED;
    }
    if (rc == DA_ERR_UNSUPPORTED) {
        rc = 0;
        int rpb;
        const size_t parts = wgrad_direct_parts((long long)N * Do * Ho, O, &rpb);
        float* partial = (float*)ws;
        hipLaunchKernelGGL(conv3_direct_wgrad_kernel, dim3((unsigned)parts), dim3(256), 0, st, in1, C1, in2, C2, dy, partial,
                           N, D, H, W, Do, Ho, Wo, Cout, stride, rpb);
        DA_LAUNCH_CHECK();
        hipLaunchKernelGGL(partial_reduce_kernel, dim3(da_grid(O, 256)), dim3(256), 0, st, partial, (int)parts, O, dw_tio);
        DA_LAUNCH_CHECK();
    }
    if (rc) return rc;
    if (dbias) {
        // the column-sum scratch sits after the largest partial region
        const size_t off = ws_bytes - da_bn_ws_bytes(0, Cout);
        rc = da_colsum(dy, (long long)N * Do * Ho * Wo, Cout, dbias, (char*)ws + (off / 256) * 256, da_bn_ws_bytes(0, Cout), stream);
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// bf16 activation storage (common.h): twins of the entries above whose activation / gradient tensors are bf16 in HBM.  `bf16_mask` has
// one bit per activation argument in signature order (bit set = that tensor is bf16).  Handled natively: all tensors bf16, stride 1, the
// matrix-core kernels of the bf16 matrix mode (da_set_matrix_mode(1)).  Everything else returns DA_ERR_UNSUPPORTED and the caller
// converts (da_cast_*) around the fp32 entry.  Weights, biases, statistics and weight gradients are always fp32.
// ---------------------------------------------------------------------------------------------------
extern "C" int da_conv3d_k3_fwd_bf16(const void* in1, int C1, const void* in2, int C2,
                                     const float* w_tio, const float* bias, void* out,
                                     int N, int D, int H, int W, int Cout, int stride, float act_slope,
                                     void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask) {
    if (!in1 || !w_tio || !out || C1 <= 0 || C2 < 0 || (C2 > 0 && !in2) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (stride != 1 && stride != 2))
        return DA_ERR_BADARG;
    const unsigned want = 1u | (C2 > 0 ? 2u : 0u) | 4u;
    if (force_direct()) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, stride)) return DA_ERR_WS_SMALL;
    {   // thin layers on the VALU kernel: fp32 network inputs -> bf16 (first layers), bf16 -> fp32 displacement field (flow conv)
        const unsigned inm = 1u | (C2 > 0 ? 2u : 0u), in_bits = bf16_mask & inm;
        if (stride == 1 && (in_bits == 0 || in_bits == inm) && !da_conv3_mfma_fwd_supported(C1, C2, Cout, stride) && da_conv3_thin_supported(C1, C2, Cout, stride) && !getenv("DA_NO_THIN")) {
            const int rc = da_conv3_thin_fwd((const float*)in1, C1, (const float*)in2, C2, w_tio, 0, bias, (float*)out, Cout, nullptr, 0, N, D, H, W, Cout, act_slope,
                                             ws, ws_bytes, da_stream(stream), in_bits != 0, (bf16_mask & 4u) != 0);
            if (rc != DA_ERR_UNSUPPORTED) return rc;
        }
    }
    if ((bf16_mask & want) != want) return DA_ERR_UNSUPPORTED;
    if (stride == 2) {
        if (!da_conv3_s2_supported(C1, C2, Cout) || s2_prefers_direct(N, D, H, W)) return DA_ERR_UNSUPPORTED;
        return da_conv3_s2_fwd((const float*)in1, C1, w_tio, bias, (float*)out, N, D, H, W, Cout, act_slope, ws, ws_bytes - da_bn_ws_bytes(0, Cout) - 4096, da_stream(stream), 1);
    }
    if (!da_conv3_mfma_fwd_supported(C1, C2, Cout, stride)) return DA_ERR_UNSUPPORTED;
    return da_conv3_mfma_fwd((const float*)in1, C1, (const float*)in2, C2, w_tio, 0, bias, (float*)out, Cout, nullptr, 0,
                             N, D, H, W, Cout, stride, act_slope, ws, ws_bytes, da_stream(stream), 0, nullptr, nullptr, nullptr, nullptr, 1);
}

extern "C" int da_conv3d_k3_fwd_bnstats_bf16(const void* in1, int C1, const void* in2, int C2,
                                             const float* w_tio, const float* bias, void* out,
                                             int N, int D, int H, int W, int Cout, int stride,
                                             double* stats_partial, int stats_capacity, int* stats_nparts,
                                             void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask) {
    if (stats_nparts) *stats_nparts = 0;
    if (!in1 || !w_tio || !out || C1 <= 0 || C2 < 0 || (C2 > 0 && !in2) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (stride != 1 && stride != 2))
        return DA_ERR_BADARG;
    const unsigned want = 1u | (C2 > 0 ? 2u : 0u) | 4u;
    if (force_direct()) return DA_ERR_UNSUPPORTED;
    if ((bf16_mask & want) != want || stride != 1 || !da_conv3_mfma_fwd_supported(C1, C2, Cout, stride))      // no fused statistics (*stats_nparts = 0): thin / stride-2 routes
        return da_conv3d_k3_fwd_bf16(in1, C1, in2, C2, w_tio, bias, out, N, D, H, W, Cout, stride, -1.f, ws, ws_bytes, stream, bf16_mask);
    if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, stride)) return DA_ERR_WS_SMALL;
    const bool stats = stats_partial && stats_capacity >= 512;
    return da_conv3_mfma_fwd((const float*)in1, C1, (const float*)in2, C2, w_tio, 0, bias, (float*)out, Cout, nullptr, 0, N, D, H, W, Cout, stride, -1.f,
                             ws, ws_bytes, da_stream(stream), 0, stats ? stats_partial : nullptr, stats ? stats_nparts : nullptr, nullptr, nullptr, 1);
}

extern "C" int da_conv3d_k3_fwd_pro_bf16(const void* in1, int C1, const float* pro1_scale, const float* pro1_shift, float pro1_slope,
                                         const void* in2, int C2, const float* pro2_scale, const float* pro2_shift, float pro2_slope,
                                         const float* w_tio, const float* bias, void* out,
                                         int N, int D, int H, int W, int Cout, float act_slope,
                                         double* stats_partial, int stats_capacity, int* stats_nparts,
                                         void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask) {
    if (stats_nparts) *stats_nparts = 0;
    if (!in1 || !w_tio || !out || C1 <= 0 || C2 < 0 || (C2 > 0 && !in2) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 ||
        (pro1_scale && !pro1_shift) || (pro2_scale && !pro2_shift))
        return DA_ERR_BADARG;
    const unsigned want = 1u | (C2 > 0 ? 2u : 0u) | 4u;
    if ((bf16_mask & want) != want || force_direct() || !da_conv3_mfma_fwd_supported(C1, C2, Cout, 1)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, 1)) return DA_ERR_WS_SMALL;
    const DaPro pro = {pro1_scale, pro1_shift, pro1_slope, pro2_scale, pro2_shift, pro2_slope};
    const bool stats = stats_partial && stats_capacity >= 512;
    return da_conv3_mfma_fwd((const float*)in1, C1, (const float*)in2, C2, w_tio, 0, bias, (float*)out, Cout, nullptr, 0, N, D, H, W, Cout, 1, stats ? -1.f : act_slope,
                             ws, ws_bytes, da_stream(stream), 0, stats ? stats_partial : nullptr, stats ? stats_nparts : nullptr, &pro, nullptr, 1);
}

extern "C" int da_conv3d_k3_wgrad_pro_bf16(const void* in1, int C1, const float* pro1_scale, const float* pro1_shift, float pro1_slope,
                                           const void* in2, int C2, const float* pro2_scale, const float* pro2_shift, float pro2_slope,
                                           const void* dy, float* dw_tio,
                                           int N, int D, int H, int W, int Cout,
                                           void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask) {
    if (!in1 || !dy || !dw_tio || C1 <= 0 || C2 < 0 || (C2 > 0 && !in2) || N <= 0 || Cout <= 0 ||
        (pro1_scale && !pro1_shift) || (pro2_scale && !pro2_shift))
        return DA_ERR_BADARG;
    const unsigned want = 1u | (C2 > 0 ? 2u : 0u) | 4u;
    if ((bf16_mask & want) != want || force_direct() || !da_conv3_mfma_wgrad_supported(C1, C2, Cout, 1)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, 1)) return DA_ERR_WS_SMALL;
    const DaPro pro = {pro1_scale, pro1_shift, pro1_slope, pro2_scale, pro2_shift, pro2_slope};
    return da_conv3_mfma_wgrad((const float*)in1, C1, (const float*)in2, C2, (const float*)dy, dw_tio, N, D, H, W, Cout, 1, ws, ws_bytes, da_stream(stream), 0, &pro, nullptr, 1);
}

extern "C" int da_conv3d_k3_dgrad_bf16(const void* dy, const float* w_tio, void* dx1, int C1, void* dx2, int C2,
                                       int N, int D, int H, int W, int Cout, int stride,
                                       void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask) {
    if (!dy || !w_tio || !dx1 || C1 <= 0 || C2 < 0 || (C2 > 0 && !dx2) || N <= 0 || Cout <= 0 || (stride != 1 && stride != 2)) return DA_ERR_BADARG;
    const int Cin = C1 + C2;
    const unsigned want = 1u | 2u | (C2 > 0 ? 4u : 0u);
    if (force_direct()) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, Cin, Cout, stride)) return DA_ERR_WS_SMALL;
    {   // data gradient of the flow conv: fp32 gradient of the displacement field -> bf16 gradients of its two inputs
        const unsigned outm = 2u | (C2 > 0 ? 4u : 0u), out_bits = bf16_mask & outm;
        if (stride == 1 && (out_bits == 0 || out_bits == outm) && !da_conv3_mfma_fwd_supported(Cout, 0, Cin, 1, C1, C2) && da_conv3_thin_supported(Cout, 0, Cin, 1)) {
            const int rc = da_conv3_thin_fwd((const float*)dy, Cout, nullptr, 0, w_tio, 1, nullptr, (float*)dx1, C1, (float*)dx2, C2, N, D, H, W, Cin, -1.f, ws, ws_bytes, da_stream(stream),
                                             (bf16_mask & 1u) != 0, out_bits != 0);
            if (rc != DA_ERR_UNSUPPORTED) return rc;
        }
    }
    if ((bf16_mask & want) != want) return DA_ERR_UNSUPPORTED;
    if (stride == 2) {
        if (!da_conv3_s2_supported(C1, C2, Cout) || s2_prefers_direct(N, D, H, W)) return DA_ERR_UNSUPPORTED;
        return da_conv3_s2_dgrad((const float*)dy, w_tio, (float*)dx1, C1, N, D, H, W, Cout, ws, ws_bytes - da_bn_ws_bytes(0, Cout) - 4096, da_stream(stream), 1);
    }
    if (!da_conv3_mfma_fwd_supported(Cout, 0, Cin, 1, C1, C2)) return DA_ERR_UNSUPPORTED;
    return da_conv3_mfma_fwd((const float*)dy, Cout, nullptr, 0, w_tio, /*w_is_flipped_tr=*/1, nullptr, (float*)dx1, C1, (float*)dx2, C2,
                             N, D, H, W, Cin, 1, -1.f, ws, ws_bytes, da_stream(stream), 0, nullptr, nullptr, nullptr, nullptr, 1);
}

extern "C" int da_conv3d_k3_wgrad_bf16(const void* in1, int C1, const void* in2, int C2, const void* dy,
                                       float* dw_tio, float* dbias,
                                       int N, int D, int H, int W, int Cout, int stride,
                                       void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask) {
    if (!in1 || !dy || !dw_tio || C1 <= 0 || C2 < 0 || (C2 > 0 && !in2) || N <= 0 || Cout <= 0 || (stride != 1 && stride != 2)) return DA_ERR_BADARG;
    const unsigned want = 1u | (C2 > 0 ? 2u : 0u) | 4u;
    if (force_direct() || dbias) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, stride)) return DA_ERR_WS_SMALL;
    {   // thin layers: first layers (fp32 network input, bf16 output gradient) and the flow conv (bf16 inputs, fp32 gradient of the field)
        const unsigned inm = 1u | (C2 > 0 ? 2u : 0u), in_bits = bf16_mask & inm;
        const bool dy_bf = (bf16_mask & 4u) != 0;
        if (stride == 1 && in_bits == 0 && dy_bf && da_conv3_fewcin_wgrad_supported(C1, C2, Cout, stride))
            return da_conv3_fewcin_wgrad((const float*)in1, C1, (const float*)in2, C2, (const float*)dy, dw_tio, N, D, H, W, Cout, ws, ws_bytes, da_stream(stream), 1);
        if (stride == 1 && in_bits == inm && !dy_bf && da_conv3_flow_wgrad_supported(C1, C2, Cout, stride) && ws_bytes >= da_conv3_flow_wgrad_ws_bytes(C1 + C2, Cout))
            return da_conv3_flow_wgrad((const float*)in1, C1, (const float*)in2, C2, (const float*)dy, dw_tio, N, D, H, W, Cout, ws, ws_bytes, da_stream(stream), 1);
    }
    if ((bf16_mask & want) != want) return DA_ERR_UNSUPPORTED;
    if (stride == 2) {
        if (!da_conv3_s2_supported(C1, C2, Cout) || s2_prefers_direct(N, D, H, W)) return DA_ERR_UNSUPPORTED;
        return da_conv3_s2_wgrad((const float*)in1, C1, (const float*)dy, dw_tio, N, D, H, W, Cout, ws, ws_bytes - da_bn_ws_bytes(0, Cout) - 4096, da_stream(stream), 1);
    }
    if (!da_conv3_mfma_wgrad_supported(C1, C2, Cout, stride)) return DA_ERR_UNSUPPORTED;
    return da_conv3_mfma_wgrad((const float*)in1, C1, (const float*)in2, C2, (const float*)dy, dw_tio, N, D, H, W, Cout, stride, ws, ws_bytes, da_stream(stream), 0, nullptr, nullptr, 1);
}

// ---- C ABI: kept packs of the folded up-sampling / native stride-2 / flow / thin kernels (conv3d_internal.h) ----------------------------------------
// The family that da_conv3d_k3_fwd / da_conv3d_k3_dgrad (up2 = 0) or da_upconv3d_k3_fwd / _dgrad (up2 = 1; D, H, W = the COARSE extents) would run for this
// shape packs its operand into `buf` (capacity `cap` bytes) and launches nothing else.  *need = bytes such a pack takes (0: the shape runs a kernel that keeps
// nothing here -- the split matrix kernels have da_conv3d_k3_prepack, the direct kernels pack nothing); buf == NULL: size query only.  *tag identifies
// (family, direction) for da_conv3d_k3_use_prepacked_any; *filled = 1 when buf was written.  KEEP IN STEP with the dispatch order of the two entries above.
extern "C" int da_conv3d_k3_prepack_any(const float* w_tio, int C1, int C2, int Cout, int dgrad, int stride, int up2, int N, int D, int H, int W,
                                        void* buf, size_t cap, size_t* need, int* tag, int* filled, void* ws, size_t ws_bytes, void* stream) {
    if (need) *need = 0;
    if (tag) *tag = 0;
    if (filled) *filled = 0;
    if (!w_tio || !need || !tag || C1 <= 0 || C2 < 0 || Cout <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return DA_ERR_BADARG;
    if (da_matrix_mode() != 2 || force_direct()) return 0;
    hipStream_t st = da_stream(stream);
    const int Cin = C1 + C2;
    float* dummy = const_cast<float*>(w_tio);                   // (never dereferenced: a fill call returns right after its pack stage)
    g_kp = KeptAny{2, w_tio, (unsigned char*)buf, buf ? cap : 0, 0, 0, 0};
    int rc = DA_ERR_UNSUPPORTED;
    if (up2) {
        if (stride == 1 && da_upconv3d_k3_supported(C1, C2, Cout))
            rc = dgrad ? da_upconv3d_k3_dgrad(dummy, w_tio, dummy, C1, C2 > 0 ? dummy : nullptr, C2, N, D, H, W, Cout, ws, ws_bytes, stream)
                       : da_upconv3d_k3_fwd(dummy, C1, C2 > 0 ? dummy : nullptr, C2, w_tio, nullptr, dummy, N, D, H, W, Cout, -1.f, ws, ws_bytes, stream);
    } else if (stride == 2) {
        if (C2 == 0 && da_conv3_s2_supported(C1, C2, Cout) && !s2_prefers_direct(N, D, H, W) && da_conv3_s2_is_native(Cin, Cout, N, D, H, W) &&
            ws_bytes >= da_conv3d_k3_ws_bytes(N, D, H, W, Cin, Cout, 2))
            rc = dgrad ? da_conv3_s2n_dgrad(dummy, w_tio, dummy, Cin, N, D, H, W, Cout, ws, ws_bytes - da_bn_ws_bytes(0, Cout) - 4096, st)
                       : da_conv3_s2n_fwd(dummy, Cin, w_tio, nullptr, dummy, N, D, H, W, Cout, -1.f, ws, ws_bytes - da_bn_ws_bytes(0, Cout) - 4096, st);
    } else if (!dgrad) {
        if (da_conv3_mfma_fwd_supported(C1, C2, Cout, 1)) rc = DA_ERR_UNSUPPORTED;
        else if (da_conv3_flowmm_supported(C1, C2, Cout, N, D, H, W)) rc = da_conv3_flowmm_fwd(dummy, C1, C2 > 0 ? dummy : nullptr, C2, w_tio, nullptr, dummy, N, D, H, W, Cout, -1.f, ws, ws_bytes, st);
        else if (da_conv3_thin_supported(C1, C2, Cout, 1) && !getenv("DA_NO_THIN"))
            rc = da_conv3_thin_fwd(dummy, C1, C2 > 0 ? dummy : nullptr, C2, w_tio, 0, nullptr, dummy, Cout, nullptr, 0, N, D, H, W, Cout, -1.f, ws, ws_bytes, st);
    } else {
        if (da_conv3_mfma_fwd_supported(Cout, 0, Cin, 1, C1, C2)) rc = DA_ERR_UNSUPPORTED;
        else if (da_conv3_flowmm_supported(C1, C2, Cout, N, D, H, W)) rc = da_conv3_flowmm_dgrad(dummy, w_tio, dummy, C1, C2 > 0 ? dummy : nullptr, C2, N, D, H, W, Cout, ws, ws_bytes, st);
        else if (da_conv3_thin_supported(Cout, 0, Cin, 1))
            rc = da_conv3_thin_fwd(dummy, Cout, nullptr, 0, w_tio, 1, nullptr, dummy, C1, C2 > 0 ? dummy : nullptr, C2, N, D, H, W, Cin, -1.f, ws, ws_bytes, st);
    }
    const KeptAny got = g_kp;
    g_kp.mode = 0;
    if (rc == DA_ERR_UNSUPPORTED || rc == DA_ERR_WS_SMALL) return 0;
    if (rc) return rc;
    *need = got.need; *tag = got.tag;
    if (filled) *filled = got.filled;
    return 0;
}
// The NEXT da_conv3d_k3_fwd / _dgrad / da_upconv3d_k3_fwd / _dgrad call of this thread on `w_tio` whose kernel family and direction carry `tag` reads its
// packed operand from buf (one call only; any other call drops the hand-over).
extern "C" void da_conv3d_k3_use_prepacked_any(const float* w_tio, const void* buf, size_t cap, int tag) {
    g_kp = KeptAny{(w_tio && buf && tag) ? 1 : 0, w_tio, (unsigned char*)const_cast<void*>(buf), cap, tag, 0, 0};
}
