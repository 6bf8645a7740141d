// Voxelwise cross-entropy family of the loss registry (lib/loss.py:739-761): 'cross_entropy' (nn.CrossEntropyLoss), 'focal'
// (FocalLoss, lib/loss.py:157-213, with its `1 - nll_loss(P)` = 1 + p_t quirk) and 'soft_cross_entropy' (SoftCrossEntropy, :100-154)
// on N x C x D x H x W logits stored channels-last: one voxel = C contiguous floats.  HBM-bound, one pass each way: a thread owns a
// voxel, keeps its C logits in registers (C <= 64), per-thread fp32 loss -> wave shuffle -> per-block double partials -> finalize.
#include "common.h"

namespace {

constexpr int kXentBlocks = 2048;
constexpr int kMaxC = 64;

__device__ __forceinline__ long long xe_label(const void* labels, int label_bytes, long long i) {
    return label_bytes == 1 ? (long long)((const unsigned char*)labels)[i] : ((const long long*)labels)[i];
}

struct XentCfg {
    int mode;            // 0 cross_entropy, 1 focal, 2 soft cross entropy
    int softmax;         // focal: P = softmax(x) (1) or x (0); soft CE: log_softmax(x) (1) or log(max(x, 1e-8)) (0)
    float gamma;         // focal
    long long ignore;    // cross_entropy ignore_index
};

// per-voxel loss value (unscaled) and, for the backward, the per-channel gradient of that value
template <bool BWD>
__device__ __forceinline__ float xent_voxel(const float* __restrict__ x, int C, long long lab, const float* __restrict__ soft,
                                            const float* __restrict__ alpha, const XentCfg cfg, float gscale, float* __restrict__ dx) {
    float v[kMaxC];
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) { v[c] = x[c]; mx = fmaxf(mx, v[c]); }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(v[c] - mx);
    const float lse = mx + logf(se);
    if (cfg.mode == 0) {
        if (lab == cfg.ignore || lab < 0 || lab >= C) {
            if (BWD) for (int c = 0; c < C; ++c) dx[c] = 0.f;
            return 0.f;
        }
        const float l = v[lab] - lse;
        if (BWD) for (int c = 0; c < C; ++c) dx[c] = gscale * (expf(v[c] - lse) - (c == lab ? 1.f : 0.f));
        return -l;
    }
    if (cfg.mode == 1) {
        if (lab < 0 || lab >= C) {                                       // (the reference's F.cross_entropy raises on such a target)
            if (BWD) for (int c = 0; c < C; ++c) dx[c] = 0.f;
            return 0.f;
        }
        const float a = alpha ? alpha[lab] : 1.f;
        const float l = v[lab] - lse;                                  // log_p = -F.cross_entropy(inputs, targets)  (loss.py:200)
        const float pt = cfg.softmax ? expf(l) : v[lab];             // probs = F.nll_loss(P, targets) = -P[t]; (1 - probs) = 1 + P[t]
        const float base = 1.f + pt;
        const float pw = powf(base, cfg.gamma);
        if (BWD) {
            // L = -a (1 + pt)^g l ;  dL/dl = -a pw (+ softmax: -a g (1+pt)^(g-1) pt l) ; dl/dx_c = [c == t] - p_c
            const float dpw = cfg.gamma * powf(base, cfg.gamma - 1.f);
            float dl = -a * pw;
            if (cfg.softmax) dl += -a * dpw * pt * l;
            for (int c = 0; c < C; ++c) dx[c] = gscale * dl * ((c == lab ? 1.f : 0.f) - expf(v[c] - lse));
            if (!cfg.softmax) dx[lab] += gscale * (-a * dpw * l);
        }
        return -a * pw * l;
    }
    // soft cross entropy
    float acc = 0.f, ts = 0.f;
    if (cfg.softmax) {
        for (int c = 0; c < C; ++c) { acc -= soft[c] * (v[c] - lse); ts += soft[c]; }
        if (BWD) for (int c = 0; c < C; ++c) dx[c] = gscale * (expf(v[c] - lse) * ts - soft[c]);
    } else {
        for (int c = 0; c < C; ++c) {
            const bool pass = v[c] >= 1e-8f;                              // pred.clamp_(min=1e-8): gradient only where not clamped
            acc -= soft[c] * logf(pass ? v[c] : 1e-8f);
            if (BWD) dx[c] = pass ? gscale * (-soft[c] / v[c]) : 0.f;
        }
    }
    return acc;
}

__global__ void xent_fwd_kernel(const float* __restrict__ x, const void* __restrict__ labels, int label_bytes, const float* __restrict__ soft,
                                const float* __restrict__ alpha, long long M, int C, XentCfg cfg, double* __restrict__ partial) {
    __shared__ double red[4];
    float acc = 0.f; float cnt = 0.f; double dacc = 0.0, dcnt = 0.0; int k = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
        const long long lab = labels ? xe_label(labels, label_bytes, i) : 0;
        acc += xent_voxel<false>(x + i * C, C, lab, soft ? soft + i * C : nullptr, alpha, cfg, 0.f, nullptr);
        if (cfg.mode == 0 && lab != cfg.ignore && lab >= 0 && lab < C) cnt += 1.f;
        if (++k == 16) { dacc += (double)acc; dcnt += (double)cnt; acc = 0.f; cnt = 0.f; k = 0; }
    }
    dacc += (double)acc; dcnt += (double)cnt;
    const double s = da_block_sum(dacc, red);
    const double n = da_block_sum(dcnt, red);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s; partial[2 * blockIdx.x + 1] = n; }
}

// reduction: 0 'mean' over the counted voxels (cross_entropy: not ignored; others: all M), 1 'sum'
__global__ void xent_finalize_kernel(const double* __restrict__ partial, int nblocks, long long M, int mode, int reduction,
                                     float* __restrict__ loss, float* __restrict__ denom) {
    __shared__ double red[4];
    double s = 0.0, n = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { s += partial[2 * i]; n += partial[2 * i + 1]; }
    s = da_block_sum(s, red);
    n = da_block_sum(n, red);
    if (threadIdx.x == 0) {
        const double d = reduction == 1 ? 1.0 : (mode == 0 ? n : (double)M);
        loss[0] = (float)(s / d);
        denom[0] = (float)d;
    }
}

__global__ void xent_bwd_kernel(const float* __restrict__ x, const void* __restrict__ labels, int label_bytes, const float* __restrict__ soft,
                                const float* __restrict__ alpha, const float* __restrict__ dloss, const float* __restrict__ denom,
                                float* __restrict__ dx, long long M, int C, XentCfg cfg) {
    const float gscale = dloss[0] / denom[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
        const long long lab = labels ? xe_label(labels, label_bytes, i) : 0;
        xent_voxel<true>(x + i * C, C, lab, soft ? soft + i * C : nullptr, alpha, cfg, gscale, dx + i * C);
    }
}

}  // namespace

extern "C" size_t da_xent_ws_bytes(void) { return da_align((size_t)kXentBlocks * 2 * sizeof(double)); }

extern "C" int da_xent_fwd(const float* logits, const void* labels, int label_bytes, const float* soft_target, const float* alpha,
                           long long M, int C, int mode, int softmax, float gamma, long long ignore_index, int reduction,
                           float* loss, float* denom, void* ws, size_t ws_bytes, void* stream) {
    if (!logits || !loss || !denom || M <= 0 || C < 1 || mode < 0 || mode > 2 || (mode < 2 && !labels) || (mode == 2 && !soft_target) ||
        (labels && label_bytes != 1 && label_bytes != 8)) return DA_ERR_BADARG;
    if (C > kMaxC) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_xent_ws_bytes()) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    const XentCfg cfg{mode, softmax, gamma, ignore_index};
    int nblocks = (int)da_cdiv(M, 256); if (nblocks > kXentBlocks) nblocks = kXentBlocks;
    hipLaunchKernelGGL(xent_fwd_kernel, dim3(nblocks), dim3(256), 0, st, logits, labels, label_bytes, soft_target, alpha, M, C, cfg, (double*)ws);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(xent_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nblocks, M, mode, reduction, loss, denom);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_xent_bwd(const float* logits, const void* labels, int label_bytes, const float* soft_target, const float* alpha,
                           const float* dloss, const float* denom, float* dlogits, long long M, int C, int mode, int softmax, float gamma,
                           long long ignore_index, void* stream) {
    if (!logits || !dloss || !denom || !dlogits || M <= 0 || C < 1 || mode < 0 || mode > 2 || (mode < 2 && !labels) || (mode == 2 && !soft_target))
        return DA_ERR_BADARG;
    if (C > kMaxC) return DA_ERR_UNSUPPORTED;
    const XentCfg cfg{mode, softmax, gamma, ignore_index};
    hipLaunchKernelGGL(xent_bwd_kernel, dim3(da_grid(M, 256)), dim3(256), 0, da_stream(stream), logits, labels, label_bytes, soft_target, alpha,
                       dloss, denom, dlogits, M, C, cfg);
    DA_LAUNCH_CHECK();
    return 0;
}
