// Voxelwise cross-entropy family of the loss registry (lib/loss.py:739-761): 'cross_entropy' (nn.CrossEntropyLoss), 'focal'
// (FocalLoss, lib/loss.py:157-213, with its `1 - nll_loss(P)` = 1 + p_t quirk) and 'soft_cross_entropy' (SoftCrossEntropy, :100-154)
// on N x C x D x H x W logits stored channels-last: one voxel = C contiguous floats.  HBM-bound, one pass each way.  C % 4 == 0: C / 4 lanes own a
// voxel (16 bytes each, reductions by xor shuffles; the second half of this file); otherwise a thread owns a voxel and keeps its C logits in
// registers (C <= 64).  Per-thread fp32 loss -> per-block double partials -> finalize.
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

// ------------------------------------------------------------------------------------------------------------------------------
// C % 4 == 0 (the registry's use: 32 classes): L = C / 4 lanes own one voxel, 16 bytes each -- a wave instruction reads 64 / L whole voxels
// (1 KiB contiguous), the logits never leave registers, and max / sum-exp / the target's logit travel across the L lanes by xor shuffles.
// (The thread-per-voxel form above reads at a C * 4-byte lane stride and indexes a private array with the label: 0.07 of HBM.)
// ------------------------------------------------------------------------------------------------------------------------------
template <int L> __device__ __forceinline__ float xe_group_max(float v) {
#pragma unroll
    for (int m = 1; m < L; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
template <int L> __device__ __forceinline__ float xe_group_sum(float v) {
#pragma unroll
    for (int m = 1; m < L; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// returns the voxel's loss value (the same on all L lanes); BWD: this lane's quad of d loss / d logits
template <int L, bool BWD>
__device__ __forceinline__ float xent_quad(const float4 q, int sub, long long lab, const float4 sq, const float* __restrict__ alpha,
                                           const XentCfg cfg, float gscale, float4& dq) {
    const int C = 4 * L;
    const float v[4] = {q.x, q.y, q.z, q.w};
    const float mx = xe_group_max<L>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
    const float se = xe_group_sum<L>(expf(v[0] - mx) + expf(v[1] - mx) + expf(v[2] - mx) + expf(v[3] - mx));
    const float lse = mx + logf(se);
    const bool own = lab >= 0 && (int)(lab >> 2) == sub;             // this lane holds the target's logit
    const int lj = (int)(lab & 3);
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    float loss = 0.f;
    if (cfg.mode == 0 || cfg.mode == 1) {
        const bool valid = lab >= 0 && lab < C && !(cfg.mode == 0 && lab == cfg.ignore);
        const float vt = xe_group_sum<L>(own ? (lj == 0 ? v[0] : lj == 1 ? v[1] : lj == 2 ? v[2] : v[3]) : 0.f);
        const float l = vt - lse;
        if (cfg.mode == 0) {
            loss = valid ? -l : 0.f;
            if (BWD && valid) {
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = gscale * (expf(v[j] - lse) - ((own && lj == j) ? 1.f : 0.f));
            }
        } else {
            const float a = (alpha && valid) ? alpha[lab] : 1.f;
            const float pt = cfg.softmax ? expf(l) : vt;
            const float base = 1.f + pt;
            const float pw = powf(base, cfg.gamma);
            loss = valid ? -a * pw * l : 0.f;
            if (BWD && valid) {
                const float dpw = cfg.gamma * powf(base, cfg.gamma - 1.f);
                float dl = -a * pw;
                if (cfg.softmax) dl += -a * dpw * pt * l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    d[j] = gscale * dl * (((own && lj == j) ? 1.f : 0.f) - expf(v[j] - lse));
                    if (!cfg.softmax && own && lj == j) d[j] += gscale * (-a * dpw * l);
                }
            }
        }
    } else {
        const float s[4] = {sq.x, sq.y, sq.z, sq.w};
        if (cfg.softmax) {
            float acc = 0.f, ts = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc -= s[j] * (v[j] - lse); ts += s[j]; }
            loss = xe_group_sum<L>(acc);
            ts = xe_group_sum<L>(ts);
            if (BWD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = gscale * (expf(v[j] - lse) * ts - s[j]);
            }
        } else {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool pass = v[j] >= 1e-8f;
                acc -= s[j] * logf(pass ? v[j] : 1e-8f);
                if (BWD) d[j] = pass ? gscale * (-s[j] / v[j]) : 0.f;
            }
            loss = xe_group_sum<L>(acc);
        }
    }
    if (BWD) dq = make_float4(d[0], d[1], d[2], d[3]);
    return loss;
}

template <int L>
__global__ void __launch_bounds__(256) xent_fwd_quad_kernel(const float* __restrict__ x, const void* __restrict__ labels, int label_bytes, const float* __restrict__ soft,
                                                            const float* __restrict__ alpha, long long M, XentCfg cfg, double* __restrict__ partial) {
    __shared__ double red[4];
    const int C = 4 * L, sub = (int)threadIdx.x & (L - 1);
    const long long vpb = 256 / L;                                     // voxels per workgroup pass
    float acc = 0.f, cnt = 0.f; double dacc = 0.0, dcnt = 0.0; int k = 0;
    const long long passes = (M + vpb - 1) / vpb;                      // every lane of a group runs the same passes (shuffles inside)
    for (long long ps = blockIdx.x; ps < passes; ps += gridDim.x) {
        const long long i = ps * vpb + (long long)(threadIdx.x / L);
        const bool in = i < M;
        const long long ii = in ? i : M - 1;
        const float4 q = reinterpret_cast<const float4*>(x + ii * C)[sub];
        const float4 sq = soft ? reinterpret_cast<const float4*>(soft + ii * C)[sub] : make_float4(0.f, 0.f, 0.f, 0.f);
        const long long lab = labels ? xe_label(labels, label_bytes, ii) : 0;
        float4 dq;
        const float l = xent_quad<L, false>(q, sub, lab, sq, alpha, cfg, 0.f, dq);
        if (in && sub == 0) {
            acc += l;
            if (cfg.mode == 0 && lab != cfg.ignore && lab >= 0 && lab < C) cnt += 1.f;
        }
        if (++k == 16) { dacc += (double)acc; dcnt += (double)cnt; acc = 0.f; cnt = 0.f; k = 0; }
    }
    dacc += (double)acc; dcnt += (double)cnt;
    const double s = da_block_sum(dacc, red);
    const double n = da_block_sum(dcnt, red);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s; partial[2 * blockIdx.x + 1] = n; }
}

template <int L>
__global__ void __launch_bounds__(256) xent_bwd_quad_kernel(const float* __restrict__ x, const void* __restrict__ labels, int label_bytes, const float* __restrict__ soft,
                                                            const float* __restrict__ alpha, const float* __restrict__ dloss, const float* __restrict__ denom,
                                                            float* __restrict__ dx, long long M, XentCfg cfg) {
    const int C = 4 * L, sub = (int)threadIdx.x & (L - 1);
    const float gscale = dloss[0] / denom[0];
    const long long vpb = 256 / L, passes = (M + vpb - 1) / vpb;
    for (long long ps = blockIdx.x; ps < passes; ps += gridDim.x) {
        const long long i = ps * vpb + (long long)(threadIdx.x / L);
        const bool in = i < M;
        const long long ii = in ? i : M - 1;
        const float4 q = reinterpret_cast<const float4*>(x + ii * C)[sub];
        const float4 sq = soft ? reinterpret_cast<const float4*>(soft + ii * C)[sub] : make_float4(0.f, 0.f, 0.f, 0.f);
        const long long lab = labels ? xe_label(labels, label_bytes, ii) : 0;
        float4 dq;
        xent_quad<L, true>(q, sub, lab, sq, alpha, cfg, gscale, dq);
        if (in) reinterpret_cast<float4*>(dx + i * C)[sub] = dq;
    }
}

static int xe_lanes(int C, const float* a, const float* b, const float* c) {      // lanes per voxel of the quad kernels, 0: thread-per-voxel form
    if (C % 4 != 0 || getenv("DA_XENT_V1")) return 0;
    const int L = C / 4;
    if (L != 1 && L != 2 && L != 4 && L != 8 && L != 16) return 0;
    if (((size_t)a | (size_t)b | (size_t)c) & 15) return 0;
    return L;
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
    const int L = xe_lanes(C, logits, soft_target, nullptr);
    if (L) {
        const long long passes = da_cdiv(M, 256 / L);
        nblocks = passes < kXentBlocks ? (int)passes : kXentBlocks;
    }
#define XE_FWD(l) case l: hipLaunchKernelGGL((xent_fwd_quad_kernel<l>), dim3(nblocks), dim3(256), 0, st, logits, labels, label_bytes, soft_target, alpha, M, cfg, (double*)ws); break;
    switch (L) {
        XE_FWD(1) XE_FWD(2) XE_FWD(4) XE_FWD(8) XE_FWD(16)
        default: hipLaunchKernelGGL(xent_fwd_kernel, dim3(nblocks), dim3(256), 0, st, logits, labels, label_bytes, soft_target, alpha, M, C, cfg, (double*)ws);
    }
#undef XE_FWD
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
    const int L = xe_lanes(C, logits, soft_target, dlogits);
#define XE_BWD(l) case l: hipLaunchKernelGGL((xent_bwd_quad_kernel<l>), dim3(da_grid(da_cdiv(M, 256 / l) * 256, 256)), dim3(256), 0, da_stream(stream), logits, labels, label_bytes, \
                                             soft_target, alpha, dloss, denom, dlogits, M, cfg); break;
    switch (L) {
        XE_BWD(1) XE_BWD(2) XE_BWD(4) XE_BWD(8) XE_BWD(16)
        default: hipLaunchKernelGGL(xent_bwd_kernel, dim3(da_grid(M, 256)), dim3(256), 0, da_stream(stream), logits, labels, label_bytes, soft_target, alpha,
                                    dloss, denom, dlogits, M, C, cfg);
    }
#undef XE_BWD
    DA_LAUNCH_CHECK();
    return 0;
}
