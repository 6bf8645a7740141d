// Stride-2 3x3x3 convolutions (the registration encoder, voxel_morph.py:46-47, modules.py:48) on the MFMA kernels.
//
// out[o] = sum_t W[t] . X[2o - 1 + t].  Split X into its 8 parity sub-volumes X_r[q] = X[2q + r]: tap t = 0 reads the
// odd sub-volume at q = o - 1, t = 1 the even one at q = o, t = 2 the odd one at q = o.  Hence the stride-2 conv is a
// STRIDE-1 conv over the space-to-depth tensor S[q][r*Cin + ci] (8*Cin channels, half resolution) whose 27-tap kernel is
// zero except for (1|2)^3 taps per parity.  The MFMA kernels take a per-chunk tap mask and skip the zero K-steps, so
// the MFMA work equals the original 27*Cin*Cout per output voxel; space-to-depth / depth-to-space are HBM-bound copies.
#include "common.h"
#include "conv3d_internal.h"

namespace {

// S[n][qd][qh][qw][r*C + c] = X[n][2qd+rz][2qh+ry][2qw+rx][c]  (zero beyond the volume), r = (rz*2+ry)*2+rx
__global__ void space_to_depth2_kernel(const float* __restrict__ x, float* __restrict__ s, int N, int D, int H, int W, int C,
                                       int Dq, int Hq, int Wq) {
    const int cq = C / 4;
    const long long total = (long long)N * Dq * Hq * Wq * 8 * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % cq); long long t = i / cq;
        const int r = (int)(t % 8); t /= 8;
        const int qw = (int)(t % Wq); t /= Wq;
        const int qh = (int)(t % Hq); t /= Hq;
        const int qd = (int)(t % Dq); const int n = (int)(t / Dq);
        const int d = 2 * qd + ((r >> 2) & 1), h = 2 * qh + ((r >> 1) & 1), w = 2 * qw + (r & 1);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d < D && h < H && w < W) v = *reinterpret_cast<const float4*>(x + ((((long long)n * D + d) * H + h) * W + w) * C + c4 * 4);
        reinterpret_cast<float4*>(s)[i] = v;
    }
}

// inverse gather: dx[n][d][h][w][c] = dS[n][d/2][h/2][w/2][r*C + c]
__global__ void depth_to_space2_kernel(const float* __restrict__ s, float* __restrict__ x, int N, int D, int H, int W, int C,
                                       int Dq, int Hq, int Wq) {
    const int cq = C / 4;
    const long long total = (long long)N * D * H * W * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % cq); long long t = i / cq;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H); t /= H;
        const int d = (int)(t % D); const int n = (int)(t / D);
        const int r = ((d & 1) * 2 + (h & 1)) * 2 + (w & 1);
        const long long q = (((long long)n * Dq + (d >> 1)) * Hq + (h >> 1)) * Wq + (w >> 1);
        reinterpret_cast<float4*>(x)[i] = *reinterpret_cast<const float4*>(s + (q * 8 + r) * C + c4 * 4);
    }
}

// W[27][Cin][Cout] -> W'[27][8*Cin][Cout]: offset index o_a (q offset o_a - 1) of parity r_a carries original tap
// (r_a = 0: o 1 <- t 1) (r_a = 1: o 0 <- t 0, o 1 <- t 2); everything else is zero.
__device__ __forceinline__ int orig_tap_axis(int r, int o) { return r == 0 ? (o == 1 ? 1 : -1) : (o == 0 ? 0 : (o == 1 ? 2 : -1)); }

__global__ void expand_weights_s2d_kernel(const float* __restrict__ w, float* __restrict__ we, int Cin, int Cout) {
    const int total = 27 * 8 * Cin * Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % Cout; int t = i / Cout;
        const int ci = t % Cin; t /= Cin;
        const int r = t % 8; const int tp = t / 8;
        const int tz = orig_tap_axis((r >> 2) & 1, tp / 9), ty = orig_tap_axis((r >> 1) & 1, (tp / 3) % 3), tx = orig_tap_axis(r & 1, tp % 3);
        we[i] = (tz >= 0 && ty >= 0 && tx >= 0) ? w[((size_t)((tz * 3 + ty) * 3 + tx) * Cin + ci) * Cout + co] : 0.f;
    }
}

// dW[t][ci][co] = dW'[o(t)][r(t)*Cin + ci][co]:  t_a = 0 -> (r 1, o 0); 1 -> (r 0, o 1); 2 -> (r 1, o 1)
__global__ void extract_wgrad_s2d_kernel(const float* __restrict__ dwe, float* __restrict__ dw, int Cin, int Cout) {
    const int total = 27 * Cin * Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % Cout; int t = i / Cout;
        const int ci = t % Cin; const int tap = t / Cin;
        const int ta[3] = {tap / 9, (tap / 3) % 3, tap % 3};
        int r = 0, o = 0;
        for (int a = 0; a < 3; ++a) { r = r * 2 + (ta[a] == 1 ? 0 : 1); o = o * 3 + (ta[a] == 0 ? 0 : 1); }
        dw[i] = dwe[((size_t)o * 8 * Cin + r * Cin + ci) * Cout + co];
    }
}

struct S2Plan { int Dq, Hq, Wq; size_t s_bytes, we_bytes, inner_bytes; };
static S2Plan s2_plan(int N, int D, int H, int W, int Cin, int Cout) {
    S2Plan p;
    p.Dq = (D + 1) / 2; p.Hq = (H + 1) / 2; p.Wq = (W + 1) / 2;
    p.s_bytes = da_align((size_t)N * p.Dq * p.Hq * p.Wq * 8 * Cin * sizeof(float));
    p.we_bytes = da_align((size_t)27 * 8 * Cin * Cout * sizeof(float));
    p.inner_bytes = da_conv3_mfma_ws_bytes(N, p.Dq, p.Hq, p.Wq, 8 * Cin, Cout, 1) + 65536;
    return p;
}

// split matrix mode: the native stride-2 kernels (conv3d_s2n.hip) take the shapes they support; DA_NO_S2N=1 keeps the masked route (A/B)
bool s2_native(int Cin, int Cout, int N, int D, int H, int W) {
    static int off = -1; if (off < 0) { const char* e = getenv("DA_NO_S2N"); off = (e && atoi(e)) ? 1 : 0; }
    return !off && da_matrix_mode() == 2 && da_conv3_s2n_supported(Cin, Cout, N, D, H, W);
}

// DA_S2D_COPY=1: the earlier route through materialised space-to-depth tensors (A/B of the fused addressing)
bool s2_fused() { static int v = -1; if (v < 0) { const char* e = getenv("DA_S2D_COPY"); v = (e && atoi(e)) ? 0 : 1; } return v == 1; }

}  // namespace

bool da_conv3_s2_supported(int C1, int C2, int Cout) { return C2 == 0 && C1 % 16 == 0 && C1 <= 32 && Cout >= 8 && Cout % 4 == 0; }

size_t da_conv3_s2_ws_bytes(int N, int D, int H, int W, int Cin, int Cout) {
    const S2Plan p = s2_plan(N, D, H, W, Cin, Cout);
    size_t b = p.s_bytes + 2 * p.we_bytes + p.inner_bytes;
    if (da_conv3_s2n_supported(Cin, Cout, N, D, H, W)) { const size_t nb = da_conv3_s2n_ws_bytes(N, D, H, W, Cin, Cout); if (nb > b) b = nb; }
    return b;
}

// ws layout: [S (space-to-depth tensor or its gradient)] [W' expanded] [dW' expanded] [inner conv scratch]
bool da_conv3_s2_is_native(int Cin, int Cout, int N, int D, int H, int W) { return s2_native(Cin, Cout, N, D, H, W); }

int da_conv3_s2_fwd(const float* in, int Cin, const float* w_tio, const float* bias, float* out,
                    int N, int D, int H, int W, int Cout, float slope, void* ws, size_t ws_bytes, hipStream_t st, int act_bf16) {
    if (act_bf16 && (da_matrix_mode() != 1 || !s2_fused())) return DA_ERR_UNSUPPORTED;      // bf16 activation storage: fused addressing, bf16 matrix mode
    if (s2_native(Cin, Cout, N, D, H, W)) return da_conv3_s2n_fwd(in, Cin, w_tio, bias, out, N, D, H, W, Cout, slope, ws, ws_bytes, st);
    const S2Plan p = s2_plan(N, D, H, W, Cin, Cout);
    if (ws_bytes < da_conv3_s2_ws_bytes(N, D, H, W, Cin, Cout)) return DA_ERR_WS_SMALL;
    float* S = (float*)ws; float* We = (float*)((char*)ws + p.s_bytes); char* inner = (char*)ws + p.s_bytes + 2 * p.we_bytes;
    const bool fused = s2_fused();
    if (!fused) {
        const long long tot = (long long)N * p.Dq * p.Hq * p.Wq * 8 * (Cin / 4);
        hipLaunchKernelGGL(space_to_depth2_kernel, dim3(da_grid(tot, 256)), dim3(256), 0, st, in, S, N, D, H, W, Cin, p.Dq, p.Hq, p.Wq);
        DA_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(expand_weights_s2d_kernel, dim3(da_grid(27 * 8 * Cin * Cout, 256, 512)), dim3(256), 0, st, w_tio, We, Cin, Cout);
    DA_LAUNCH_CHECK();
    const DaS2dFuse f = {D, H, W, 1, 0};
    return da_conv3_mfma_fwd(fused ? in : S, 8 * Cin, nullptr, 0, We, 0, bias, out, Cout, nullptr, 0, N, p.Dq, p.Hq, p.Wq, Cout, 1, slope,
                             inner, p.inner_bytes, st, Cin, nullptr, nullptr, nullptr, fused ? &f : nullptr, act_bf16);
}

int da_conv3_s2_dgrad(const float* dy, const float* w_tio, float* dx, int Cin, int N, int D, int H, int W, int Cout,
                      void* ws, size_t ws_bytes, hipStream_t st, int act_bf16) {
    if (act_bf16 && (da_matrix_mode() != 1 || !s2_fused())) return DA_ERR_UNSUPPORTED;
    if (s2_native(Cin, Cout, N, D, H, W)) return da_conv3_s2n_dgrad(dy, w_tio, dx, Cin, N, D, H, W, Cout, ws, ws_bytes, st);
    const S2Plan p = s2_plan(N, D, H, W, Cin, Cout);
    if (ws_bytes < da_conv3_s2_ws_bytes(N, D, H, W, Cin, Cout)) return DA_ERR_WS_SMALL;
    float* dS = (float*)ws; float* We = (float*)((char*)ws + p.s_bytes); char* inner = (char*)ws + p.s_bytes + 2 * p.we_bytes;
    hipLaunchKernelGGL(expand_weights_s2d_kernel, dim3(da_grid(27 * 8 * Cin * Cout, 256, 512)), dim3(256), 0, st, w_tio, We, Cin, Cout);
    DA_LAUNCH_CHECK();
    // dS = conv(dY, flip/transpose(W')) : logical Cin = Cout, logical Cout = 8*Cin
    const bool fused = s2_fused();
    const DaS2dFuse f = {D, H, W, 0, 1};
    int rc = da_conv3_mfma_fwd(dy, Cout, nullptr, 0, We, 1, nullptr, fused ? dx : dS, 8 * Cin, nullptr, 0, N, p.Dq, p.Hq, p.Wq, 8 * Cin, 1, -1.f,
                               inner, p.inner_bytes, st, Cin, nullptr, nullptr, nullptr, fused ? &f : nullptr, act_bf16);
    if (rc || fused) return rc;
    const long long tot = (long long)N * D * H * W * (Cin / 4);
    hipLaunchKernelGGL(depth_to_space2_kernel, dim3(da_grid(tot, 256)), dim3(256), 0, st, dS, dx, N, D, H, W, Cin, p.Dq, p.Hq, p.Wq);
    DA_LAUNCH_CHECK();
    return 0;
}

int da_conv3_s2_wgrad(const float* in, int Cin, const float* dy, float* dw_tio, int N, int D, int H, int W, int Cout,
                      void* ws, size_t ws_bytes, hipStream_t st, int act_bf16) {
    if (act_bf16 && (da_matrix_mode() != 1 || !s2_fused())) return DA_ERR_UNSUPPORTED;
    if (s2_native(Cin, Cout, N, D, H, W)) return da_conv3_s2n_wgrad(in, Cin, dy, dw_tio, N, D, H, W, Cout, ws, ws_bytes, st);
    const S2Plan p = s2_plan(N, D, H, W, Cin, Cout);
    if (ws_bytes < da_conv3_s2_ws_bytes(N, D, H, W, Cin, Cout)) return DA_ERR_WS_SMALL;
    float* S = (float*)ws; float* dWe = (float*)((char*)ws + p.s_bytes + p.we_bytes); char* inner = (char*)ws + p.s_bytes + 2 * p.we_bytes;
    const bool fused = s2_fused();
    if (!fused) {
        const long long tot = (long long)N * p.Dq * p.Hq * p.Wq * 8 * (Cin / 4);
        hipLaunchKernelGGL(space_to_depth2_kernel, dim3(da_grid(tot, 256)), dim3(256), 0, st, in, S, N, D, H, W, Cin, p.Dq, p.Hq, p.Wq);
        DA_LAUNCH_CHECK();
    }
    const DaS2dFuse f = {D, H, W, 1, 0};
    int rc = da_conv3_mfma_wgrad(fused ? in : S, 8 * Cin, nullptr, 0, dy, dWe, N, p.Dq, p.Hq, p.Wq, Cout, 1, inner, p.inner_bytes, st, Cin, nullptr, fused ? &f : nullptr, act_bf16);
    if (rc) return rc;
    hipLaunchKernelGGL(extract_wgrad_s2d_kernel, dim3(da_grid(27 * Cin * Cout, 256, 512)), dim3(256), 0, st, dWe, dw_tio, Cin, Cout);
    DA_LAUNCH_CHECK();
    return 0;
}
