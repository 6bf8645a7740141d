// Registration losses of SURVEY.md row f2 (gfx950): VoxelMorphLNCC (lib/loss.py:589-617, registry name 'lncc') and gradientLoss
// (lib/loss.py:625-671, 'gradient').  Both are HBM-bound stencil + reduction passes; reductions go through per-block double
// partials and a one-block finalize (deterministic, no float atomics).
#include "common.h"

namespace {

constexpr int kBlocks = 4096;

__global__ void sum_partials_kernel(const double* __restrict__ partial, int count, double scale, double offset, float* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += blockDim.x) s += partial[i];
    s = da_block_sum(s, red);
    if (threadIdx.x == 0) out[0] = (float)(offset + scale * s);
}

// ------------------------------------------------------------------------------------------------
// LNCC.  The reference runs five F^3 all-ones convolutions (I, J, I^2, J^2, IJ; valid padding) and then, per window,
//   cross = S_IJ - Ibar S_J - Jbar S_I + Ibar Jbar n ;  Ivar = S_II - 2 Ibar S_I + Ibar^2 n ;  Jvar likewise ;
//   cc = cross^2 / (Ivar Jvar + eps) ;  loss = 1 - mean(cc).
// An all-ones F^3 filter is separable: three 1-D box sums (W, H, D).  Fields are planar [5][N][z][y][x] fp32.
// ------------------------------------------------------------------------------------------------
// pass 1: products + box sum along W:  t1[k][n][z][y][xo], xo < Wo = W - F + 1
__global__ void lncc_boxw_kernel(const float* __restrict__ I, const float* __restrict__ J, float* __restrict__ t1,
                                 long long rows, int W, int Wo, int F, long long plane, int dil, int stride) {
    const long long total = rows * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo); const long long r = i / Wo;
        const float* a = I + r * W + (long long)xo * stride; const float* b = J + r * W + (long long)xo * stride;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
        for (int t = 0; t < F; ++t) {
            const float x = a[t * dil], y = b[t * dil];
            s0 += x; s1 += y; s2 += x * x; s3 += y * y; s4 += x * y;
        }
        t1[i] = s0; t1[plane + i] = s1; t1[2 * plane + i] = s2; t1[3 * plane + i] = s3; t1[4 * plane + i] = s4;
    }
}

// generic 1-D box sum of K planar fields along one axis of a [M][A][B] view (axis length A, inner stride B):
// transposed == 0 (forward, valid):  out[m][ao][b] = sum_{t<F} in[m][ao*stride + t*dil][b],            ao < Ao
// transposed != 0 (its adjoint):     out[m][a][b]  = sum_{t<F, (a - t*dil) % stride == 0} in[m][(a - t*dil)/stride][b],  a < Ao
__global__ void box_axis_kernel(const float* __restrict__ in, float* __restrict__ out, long long M, int A, int Ao, long long B, int F,
                                int transposed, int dil, int stride) {
    const long long total = M * Ao * B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i % B; long long r = i / B;
        const int ao = (int)(r % Ao); const long long m = r / Ao;
        const float* src = in + (m * A) * B + b;
        float s = 0.f;
        for (int t = 0; t < F; ++t) {
            int a;
            if (!transposed) a = ao * stride + t * dil;
            else { const int rem = ao - t * dil; if (rem < 0 || rem % stride) continue; a = rem / stride; }
            if (a < A) s += src[(long long)a * B];
        }
        out[i] = s;
    }
}

__device__ __forceinline__ void lncc_terms(float sI, float sJ, float sII, float sJJ, float sIJ, float n, float eps,
                                           float& cross, float& ivar, float& jvar, float& den) {
    // same operation order as lib/loss.py:606-612 (fp32)
    const float im = sI / n, jm = sJ / n;
    cross = sIJ - im * sJ - jm * sI + im * jm * n;
    ivar = sII - 2.f * im * sI + im * im * n;
    jvar = sJJ - 2.f * jm * sJ + jm * jm * n;
    den = ivar * jvar + eps;
}

// pass 3: box sum along D of the 5 fields (in: [5][N][D][Ho*Wo]) fused with cc and its reduction; writes the window sums
// (kept for the backward pass): sums[k][n][zo][yo*xo]
__global__ void lncc_boxd_cc_kernel(const float* __restrict__ t2, float* __restrict__ sums, int N, int D, int Do, long long HW,
                                    int F, float n, float eps, double* __restrict__ partial, int dil, int stride) {
    __shared__ double red[4];
    const long long pin = (long long)N * D * HW, pout = (long long)N * Do * HW;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pout; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i % HW; long long r = i / HW;
        const int zo = (int)(r % Do); const long long nn = r / Do;
        const float* src = t2 + (nn * D + (long long)zo * stride) * HW + b;
        float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < F; ++t)
#pragma unroll
            for (int k = 0; k < 5; ++k) s[k] += src[k * pin + (long long)t * dil * HW];
#pragma unroll
        for (int k = 0; k < 5; ++k) sums[k * pout + i] = s[k];
        float cross, ivar, jvar, den;
        lncc_terms(s[0], s[1], s[2], s[3], s[4], n, eps, cross, ivar, jvar, den);
        acc += (double)((cross * cross) / den);
    }
    const double tot = da_block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// backward, per window: A = d cc / d cross, B = d cc / d Ivar, C = d cc / d Jvar (times -dloss / M), and the fields whose
// transposed box sums give the voxel gradients:
//   dL/dI_p = J_p [A] - [A Jbar] + 2 I_p [B] - 2 [B Ibar] ;  dL/dJ_p = I_p [A] - [A Ibar] + 2 J_p [C] - 2 [C Jbar]
// ([.] = sum over the windows containing p).  G: [7][N][Do][Ho][Wo] = A, A Ibar, A Jbar, B, B Ibar, C, C Jbar
__global__ void lncc_bwd_fields_kernel(const float* __restrict__ sums, float* __restrict__ G, long long P, float n, float eps,
                                       const float* __restrict__ dloss, float inv_m) {
    const float gs = -dloss[0] * inv_m;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
        const float sI = sums[i], sJ = sums[P + i];
        float cross, ivar, jvar, den;
        lncc_terms(sI, sJ, sums[2 * P + i], sums[3 * P + i], sums[4 * P + i], n, eps, cross, ivar, jvar, den);
        const float im = sI / n, jm = sJ / n;
        const float A = gs * 2.f * cross / den;
        const float q = gs * -(cross * cross) / (den * den);
        const float B = q * jvar, C = q * ivar;
        G[i] = A; G[P + i] = A * im; G[2 * P + i] = A * jm; G[3 * P + i] = B; G[4 * P + i] = B * im; G[5 * P + i] = C; G[6 * P + i] = C * jm;
    }
}

// last backward pass: full box sum along W of the 7 fields (in: [7][rows][Wo]) fused with the combine
__global__ void lncc_bwd_boxw_combine_kernel(const float* __restrict__ t, const float* __restrict__ I, const float* __restrict__ J,
                                             float* __restrict__ dI, float* __restrict__ dJ, long long rows, int W, int Wo, int F,
                                             int dil, int stride) {
    const long long total = rows * W, plane = rows * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); const long long r = i / W;
        float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int tt = 0; tt < F; ++tt) {
            const int rem = x - tt * dil;
            if (rem < 0 || rem % stride) continue;
            const int xo = rem / stride;
            if (xo < Wo) {
#pragma unroll
                for (int k = 0; k < 7; ++k) s[k] += t[k * plane + r * Wo + xo];
            }
        }
        const float a = I[i], b = J[i];
        if (dI) dI[i] = b * s[0] - s[2] + 2.f * a * s[3] - 2.f * s[4];
        if (dJ) dJ[i] = a * s[0] - s[1] + 2.f * b * s[5] - 2.f * s[6];
    }
}

// ------------------------------------------------------------------------------------------------
// gradientLoss (lib/loss.py:625-671) on disp [N][D][H][W][3] (NDHWC).  Central differences WITHOUT the 1/2h and with the
// reference's sign quirk: along D it is u(+1) - u(-1), along H and W it is u(+1) + u(-1) (lib/loss.py:659-663).
// L2: mean over the axis-interior voxels of d^2, times (spatial_dims[c] spacing[c] / spacing[k])^2 where the 3-vector is
// broadcast over the CHANNEL axis (as in BendingEnergyLoss), then mean over (N, 3), then the three axes averaged.
// L1: plain mean of |d| (the weights only appear in the L2 branch).
// ------------------------------------------------------------------------------------------------
struct GradK { float k[3][3]; int l2; };     // k[c][axis]

static GradK gradloss_coeffs(int N, int D, int H, int W, const float* spacing3, int normalize, int norm) {
    float sp[3] = {1.f, 1.f, 1.f};
    if (spacing3) { sp[0] = spacing3[0]; sp[1] = spacing3[1]; sp[2] = spacing3[2]; }
    if (normalize) { const float m = fminf(sp[0], fminf(sp[1], sp[2])); sp[0] /= m; sp[1] /= m; sp[2] /= m; }
    float dims[3] = {(float)D, (float)H, (float)W};
    if (normalize) { const float m = fminf(dims[0], fminf(dims[1], dims[2])); dims[0] /= m; dims[1] /= m; dims[2] /= m; }
    const double cnt[3] = {(double)(D - 2) * H * W, (double)D * (H - 2) * W, (double)D * H * (W - 2)};
    GradK K; K.l2 = (norm == 2);
    for (int c = 0; c < 3; ++c)
        for (int k = 0; k < 3; ++k) {
            const float w = K.l2 ? dims[c] * sp[c] / sp[k] : 1.f;
            K.k[c][k] = (float)((double)(w * w) / (cnt[k] * 3.0 * (double)N * 3.0));
        }
    return K;
}

#define GU(dd, hh, ww) u[((((long long)(dd)) * H + (hh)) * W + (ww)) * 3 + c]

__global__ void gradloss_partial_kernel(const float* __restrict__ disp, int D, int H, int W, GradK K, double* __restrict__ partial) {
    __shared__ double red[4];
    const int n = blockIdx.y;
    const float* u = disp + (long long)n * D * H * W * 3;
    const long long total = (long long)D * H * W * 3;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3); long long r = i / 3;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const int d = (int)(r / H);
        float v = 0.f;
        if (d >= 1 && d < D - 1) { const float t = fabsf(GU(d + 1, h, w) - GU(d - 1, h, w)); v += K.k[c][0] * (K.l2 ? t * t : t); }
        if (h >= 1 && h < H - 1) { const float t = fabsf(GU(d, h + 1, w) + GU(d, h - 1, w)); v += K.k[c][1] * (K.l2 ? t * t : t); }
        if (w >= 1 && w < W - 1) { const float t = fabsf(GU(d, h, w + 1) + GU(d, h, w - 1)); v += K.k[c][2] * (K.l2 ? t * t : t); }
        acc += (double)v;
    }
    const double tot = da_block_sum(acc, red);
    if (threadIdx.x == 0) partial[(size_t)n * gridDim.x + blockIdx.x] = tot;
}

// gather form: voxel p gets a term from the stencil centred one step below it and one step above it, per axis
__global__ void gradloss_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ dloss, float* __restrict__ d_disp,
                                    int D, int H, int W, GradK K) {
    const int n = blockIdx.y;
    const float* u = disp + (long long)n * D * H * W * 3;
    float* du = d_disp + (long long)n * D * H * W * 3;
    const long long total = (long long)D * H * W * 3;
    const float gl = dloss[0];
    auto fp = [&](float t) -> float { return K.l2 ? 2.f * t : (t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f)); };   // d f(|t|) / dt
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3); long long r = i / 3;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const int d = (int)(r / H);
        float g = 0.f;
        // axis D: t(q) = u(q+1) - u(q-1); p = q+1 -> +f'(t), p = q-1 -> -f'(t)
        if (d - 1 >= 1 && d - 1 < D - 1) g += K.k[c][0] * fp(GU(d, h, w) - GU(d - 2, h, w));
        if (d + 1 >= 1 && d + 1 < D - 1) g -= K.k[c][0] * fp(GU(d + 2, h, w) - GU(d, h, w));
        // axes H, W: t(q) = u(q+1) + u(q-1); both neighbours get +f'(t)
        if (h - 1 >= 1 && h - 1 < H - 1) g += K.k[c][1] * fp(GU(d, h, w) + GU(d, h - 2, w));
        if (h + 1 >= 1 && h + 1 < H - 1) g += K.k[c][1] * fp(GU(d, h + 2, w) + GU(d, h, w));
        if (w - 1 >= 1 && w - 1 < W - 1) g += K.k[c][2] * fp(GU(d, h, w) + GU(d, h, w - 2));
        if (w + 1 >= 1 && w + 1 < W - 1) g += K.k[c][2] * fp(GU(d, h, w + 2) + GU(d, h, w));
        du[i] = gl * g;
    }
}

}  // namespace

// ================================================================================================
static inline int lncc_out(int L, int F, int dil, int stride) { const int span = dil * (F - 1) + 1; return L < span ? 0 : (L - span) / stride + 1; }

extern "C" size_t da_lncc_ws_bytes(int N, int D, int H, int W, int F, int dil, int stride) {
    if (F < 1 || dil < 1 || stride < 1) return 0;
    const size_t Wo = lncc_out(W, F, dil, stride), Ho = lncc_out(H, F, dil, stride), Do = lncc_out(D, F, dil, stride);
    if (!Wo || !Ho || !Do) return 0;
    // forward: t1 [5][N][D][H][Wo], t2 [5][N][D][Ho][Wo], partials; backward: G [7][N][Do][Ho][Wo], [7][N][D][Ho][Wo], [7][N][D][H][Wo]
    const size_t fwd = da_align((size_t)5 * N * D * H * Wo * 4) + da_align((size_t)5 * N * D * Ho * Wo * 4) + da_align((size_t)kBlocks * 8);
    const size_t bwd = da_align((size_t)7 * N * Do * Ho * Wo * 4) + da_align((size_t)7 * N * D * Ho * Wo * 4) + da_align((size_t)7 * N * D * H * Wo * 4);
    return fwd > bwd ? fwd : bwd;
}

extern "C" int da_lncc_fwd(const float* I, const float* J, int N, int D, int H, int W, int F, int dil, int stride, float eps,
                           float* loss, float* sums, void* ws, size_t ws_bytes, void* stream) {
    if (!I || !J || !loss || !sums || N <= 0 || F < 1 || dil < 1 || stride < 1) return DA_ERR_BADARG;
    const int Wo = lncc_out(W, F, dil, stride), Ho = lncc_out(H, F, dil, stride), Do = lncc_out(D, F, dil, stride);
    if (!Wo || !Ho || !Do) return DA_ERR_BADARG;
    if (ws_bytes < da_lncc_ws_bytes(N, D, H, W, F, dil, stride)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    float* t1 = (float*)ws;
    float* t2 = (float*)((char*)ws + da_align((size_t)5 * N * D * H * Wo * 4));
    double* partial = (double*)((char*)t2 + da_align((size_t)5 * N * D * Ho * Wo * 4));
    const long long rows = (long long)N * D * H;
    hipLaunchKernelGGL(lncc_boxw_kernel, dim3(da_grid(rows * Wo, 256)), dim3(256), 0, st, I, J, t1, rows, W, Wo, F, rows * Wo, dil, stride);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(box_axis_kernel, dim3(da_grid((long long)5 * N * D * Ho * Wo, 256)), dim3(256), 0, st, t1, t2, (long long)5 * N * D, H, Ho, (long long)Wo, F, 0, dil, stride);
    DA_LAUNCH_CHECK();
    const long long pout = (long long)N * Do * Ho * Wo;
    int nblocks = (int)da_cdiv(pout, 256); if (nblocks > kBlocks) nblocks = kBlocks;
    hipLaunchKernelGGL(lncc_boxd_cc_kernel, dim3(nblocks), dim3(256), 0, st, t2, sums, N, D, Do, (long long)Ho * Wo, F, (float)((double)F * F * F), eps, partial, dil, stride);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, nblocks, -1.0 / (double)pout, 1.0, loss);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_lncc_bwd(const float* I, const float* J, const float* sums, const float* dloss, float* dI, float* dJ,
                           int N, int D, int H, int W, int F, int dil, int stride, float eps, void* ws, size_t ws_bytes, void* stream) {
    if (!I || !J || !sums || !dloss || N <= 0 || F < 1 || dil < 1 || stride < 1) return DA_ERR_BADARG;
    if (!dI && !dJ) return 0;
    const int Wo = lncc_out(W, F, dil, stride), Ho = lncc_out(H, F, dil, stride), Do = lncc_out(D, F, dil, stride);
    if (!Wo || !Ho || !Do) return DA_ERR_BADARG;
    if (ws_bytes < da_lncc_ws_bytes(N, D, H, W, F, dil, stride)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    float* G = (float*)ws;
    float* g1 = (float*)((char*)ws + da_align((size_t)7 * N * Do * Ho * Wo * 4));
    float* g2 = (float*)((char*)g1 + da_align((size_t)7 * N * D * Ho * Wo * 4));
    const long long P = (long long)N * Do * Ho * Wo;
    hipLaunchKernelGGL(lncc_bwd_fields_kernel, dim3(da_grid(P, 256)), dim3(256), 0, st, sums, G, P, (float)((double)F * F * F), eps, dloss, (float)(1.0 / (double)P));
    DA_LAUNCH_CHECK();
    // adjoint of the box filter along D, then H, then W (+ combine)
    hipLaunchKernelGGL(box_axis_kernel, dim3(da_grid((long long)7 * N * D * Ho * Wo, 256)), dim3(256), 0, st, G, g1, (long long)7 * N, Do, D, (long long)Ho * Wo, F, 1, dil, stride);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(box_axis_kernel, dim3(da_grid((long long)7 * N * D * H * Wo, 256)), dim3(256), 0, st, g1, g2, (long long)7 * N * D, Ho, H, (long long)Wo, F, 1, dil, stride);
    DA_LAUNCH_CHECK();
    const long long rows = (long long)N * D * H;
    hipLaunchKernelGGL(lncc_bwd_boxw_combine_kernel, dim3(da_grid(rows * W, 256)), dim3(256), 0, st, g2, I, J, dI, dJ, rows, W, Wo, F, dil, stride);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t da_gradloss_ws_bytes(int N, int D, int H, int W) {
    (void)D; (void)H; (void)W;
    return da_align((size_t)N * kBlocks * sizeof(double));
}

extern "C" int da_gradloss_fwd(const float* disp, int N, int D, int H, int W, const float* spacing3, int normalize, int norm,
                               float* loss, void* ws, size_t ws_bytes, void* stream) {
    if (!disp || !loss || N <= 0 || D < 3 || H < 3 || W < 3 || (norm != 1 && norm != 2)) return DA_ERR_BADARG;
    if (ws_bytes < da_gradloss_ws_bytes(N, D, H, W)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    const GradK K = gradloss_coeffs(N, D, H, W, spacing3, normalize, norm);
    const long long total = (long long)D * H * W * 3;
    int nblocks = (int)da_cdiv(total, 256 * 2); if (nblocks > kBlocks) nblocks = kBlocks; if (nblocks < 1) nblocks = 1;
    hipLaunchKernelGGL(gradloss_partial_kernel, dim3(nblocks, N), dim3(256), 0, st, disp, D, H, W, K, (double*)ws);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nblocks * N, 1.0, 0.0, loss);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_gradloss_bwd(const float* disp, const float* dloss, float* d_disp, int N, int D, int H, int W,
                               const float* spacing3, int normalize, int norm, void* stream) {
    if (!disp || !dloss || !d_disp || N <= 0 || D < 3 || H < 3 || W < 3 || (norm != 1 && norm != 2)) return DA_ERR_BADARG;
    const GradK K = gradloss_coeffs(N, D, H, W, spacing3, normalize, norm);
    const long long total = (long long)D * H * W * 3;
    hipLaunchKernelGGL(gradloss_bwd_kernel, dim3(da_grid(total, 256, 4096), N), dim3(256), 0, da_stream(stream), disp, dloss, d_disp, D, H, W, K);
    DA_LAUNCH_CHECK();
    return 0;
}
