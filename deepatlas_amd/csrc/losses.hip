// Voxelwise losses and metrics: fused softmax+one-hot+Dice, NCC, bending energy, eval argmax+overlap counts.
// Rows a11-a13, a15 of SURVEY.md §8: DiceLossMultiClass (lib/loss.py:410-476) + mask_to_one_hot
// (lib/transforms.py:675-689), NormalizedCrossCorrelationLoss (lib/loss.py:493-501), BendingEnergyLoss
// (lib/loss.py:687-730), eval Dice (models/segmentation.py:188-194, lib/evalMetrics.py:58-68).
// All are HBM-bound single passes: per-lane fp32 partials -> wave shuffles -> per-block double partials ->
// a one-block finalize launch (deterministic; no float atomics).
#include "common.h"

namespace {

constexpr int kBlocks = 512;     // partial blocks per sample
constexpr int kBendBlocks = 4096; // bending forward: 25-point stencil per element, latency-bound without occupancy
constexpr int kDiceBlocks = 2048; // Dice forward: 8 workgroups per CU per sample (softmax latency needs the occupancy)

__device__ __forceinline__ long long load_label(const void* labels, int label_bytes, long long i) {
    return label_bytes == 1 ? (long long)((const unsigned char*)labels)[i] : ((const long long*)labels)[i];
}

// ------------------------------------------------------------------------------------------------
// softmax helpers: LPV lanes (power of two, <= 64) own one voxel, 4 channels each.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void group_softmax4(float (&x)[4], int lpv) {
    float m = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
    for (int o = 1; o < lpv; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = expf(x[j] - m); s += x[j]; }
    for (int o = 1; o < lpv; o <<= 1) s += __shfl_xor(s, o);
    const float inv = 1.f / s;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] *= inv;
}

// ------------------------------------------------------------------------------------------------
// Dice forward partial sums.  partial[n][block][3][C] doubles = (I, S, T)
// ------------------------------------------------------------------------------------------------
__global__ void dice_partial_vec_kernel(const float* __restrict__ src, const void* __restrict__ labels, int label_bytes,
                                        const float* __restrict__ soft, long long V, int C, int lpv, int softmax,
                                        double* __restrict__ partial, float* __restrict__ prob_out = nullptr) {
    // prob_out (with softmax): the probabilities are also WRITTEN ([N][V][C]) -- the joint step's segmentation phase needs softmax(logits)
    // for the warp and the supervised Dice sums of the same logits; one pass over the logits instead of two (da_softmax_dice_fwd)
    extern __shared__ float shf[];   // [3][slots][C]
    const int n = blockIdx.y;
    const int slots = blockDim.x / lpv;
    const int q = threadIdx.x % lpv, s = threadIdx.x / lpv;
    float aI[4] = {0, 0, 0, 0}, aS[4] = {0, 0, 0, 0}, aT[4] = {0, 0, 0, 0};
    const long long vpb = da_cdiv(V, (long long)gridDim.x);
    const long long v0 = (long long)blockIdx.x * vpb;
    long long v1 = v0 + vpb; if (v1 > V) v1 = V;
    // every lane of a voxel group runs the same trip count (v depends on s only).  Two voxels per iteration: both 16-byte loads
    // (and label bytes) are in flight before the first softmax's shuffles, which doubles the bytes in flight per wave.
    auto accumulate = [&](float4 a, const float* tt, long long row) {
        float p[4] = {a.x, a.y, a.z, a.w};
        if (softmax) group_softmax4(p, lpv);
        if (prob_out) *reinterpret_cast<float4*>(prob_out + row * C + q * 4) = make_float4(p[0], p[1], p[2], p[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { aI[j] += p[j] * tt[j]; aS[j] += p[j]; aT[j] += tt[j]; }
    };
    auto target = [&](long long row, float* tt) {
        if (soft) {
            const float4 b = *reinterpret_cast<const float4*>(soft + row * C + q * 4);
            tt[0] = b.x; tt[1] = b.y; tt[2] = b.z; tt[3] = b.w;
        } else {
            const int lab = (int)load_label(labels, label_bytes, row) - q * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) tt[j] = (lab == j) ? 1.f : 0.f;
        }
    };
    long long v = v0 + s;
    for (; v + slots < v1; v += 2 * slots) {
        const long long r0 = (long long)n * V + v, r1 = r0 + slots;
        const float4 a0 = *reinterpret_cast<const float4*>(src + r0 * C + q * 4);
        const float4 a1 = *reinterpret_cast<const float4*>(src + r1 * C + q * 4);
        float t0[4], t1[4];
        target(r0, t0); target(r1, t1);
        accumulate(a0, t0, r0); accumulate(a1, t1, r1);
    }
    for (; v < v1; v += slots) {
        const long long r0 = (long long)n * V + v;
        const float4 a0 = *reinterpret_cast<const float4*>(src + r0 * C + q * 4);
        float t0[4];
        target(r0, t0);
        accumulate(a0, t0, r0);
    }
    float* sI = shf; float* sS = shf + (size_t)slots * C; float* sT = shf + (size_t)2 * slots * C;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sI[s * C + q * 4 + j] = aI[j]; sS[s * C + q * 4 + j] = aS[j]; sT[s * C + q * 4 + j] = aT[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double tI = 0, tS = 0, tT = 0;
        for (int k = 0; k < slots; ++k) { tI += sI[k * C + c]; tS += sS[k * C + c]; tT += sT[k * C + c]; }
        double* o = partial + (((size_t)n * gridDim.x + blockIdx.x) * 3) * C;
        o[c] = tI; o[C + c] = tS; o[2 * C + c] = tT;
    }
}

// generic channel count: one thread per voxel, channels looped, per-channel wave reductions.
__global__ void dice_partial_gen_kernel(const float* __restrict__ src, const void* __restrict__ labels, int label_bytes,
                                        const float* __restrict__ soft, long long V, int C, int softmax,
                                        double* __restrict__ partial) {
    extern __shared__ float shf[];   // [4 waves][3][C]: every wave owns its accumulators, combined in wave order (deterministic)
    const int n = blockIdx.y;
    for (int c = threadIdx.x; c < 4 * 3 * C; c += blockDim.x) shf[c] = 0.f;
    __syncthreads();
    float* mine = shf + (threadIdx.x >> 6) * 3 * C;
    const long long vpb = da_cdiv(V, (long long)gridDim.x);
    const long long v0 = (long long)blockIdx.x * vpb;
    long long v1 = v0 + vpb; if (v1 > V) v1 = V;
    const long long span = da_cdiv(v1 - v0, (long long)blockDim.x) * blockDim.x;
    for (long long k = threadIdx.x; k < span; k += blockDim.x) {
        const long long v = v0 + k;
        const bool live = v < v1;
        const long long row = (long long)n * V + (live ? v : v0);
        float m = -INFINITY, ssum = 1.f;
        if (softmax) {
            for (int c = 0; c < C; ++c) m = fmaxf(m, src[row * C + c]);
            ssum = 0.f;
            for (int c = 0; c < C; ++c) ssum += expf(src[row * C + c] - m);
        }
        const int lab = soft ? -1 : (int)load_label(labels, label_bytes, row);
        for (int c = 0; c < C; ++c) {
            float p = src[row * C + c];
            if (softmax) p = expf(p - m) / ssum;
            float t = soft ? soft[row * C + c] : (lab == c ? 1.f : 0.f);
            if (!live) { p = 0.f; t = 0.f; }
            const float wI = da_wave_sum(p * t), wS = da_wave_sum(p), wT = da_wave_sum(t);
            if ((threadIdx.x & 63) == 0) { mine[c] += wI; mine[C + c] += wS; mine[2 * C + c] += wT; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double* o = partial + (((size_t)n * gridDim.x + blockIdx.x) * 3) * C;
        double a = 0.0, b = 0.0, t = 0.0;
        for (int w = 0; w < 4; ++w) { a += shf[w * 3 * C + c]; b += shf[w * 3 * C + C + c]; t += shf[w * 3 * C + 2 * C + c]; }
        o[c] = a; o[C + c] = b; o[2 * C + c] = t;
    }
}

// one block; sums the partials, then thread 0 does the (N x C) epilogue in fp32 like the reference.
// coef[0][n][c] = A (multiplies the target), coef[1][n][c] = B:  dL/dp = A*t + B
// partial viewed as [N][nblocks][R]: one wave per (n, r) column sums its nblocks entries and leaves the total in slot b = 0
// (only that wave touches that column, so in place).  The single-block finalize kernels then read one value per column
// instead of walking nblocks entries serially per thread (that walk was 70-140 us of pure latency).
__global__ void colreduce_inplace_kernel(double* __restrict__ partial, int N, int nblocks, int R) {
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (wave >= N * R) return;
    const int n = wave / R, r = wave % R;
    double* col = partial + (size_t)n * nblocks * R + r;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += col[(size_t)b * R];
    s = da_wave_sum(s);
    if (lane == 0) col[0] = s;
}

__global__ void dice_finalize_kernel(const double* __restrict__ partial, int nblocks, int nsum, int N, int C,
                                     int weight_type, int no_bg, float eps, float* __restrict__ loss,
                                     float* __restrict__ coef, float* __restrict__ isc /* [3][N][C] scratch */) {
    const int NC = N * C;
    for (int i = threadIdx.x; i < 3 * NC; i += blockDim.x) {
        const int k = i / NC, nc = i % NC, n = nc / C, c = nc % C;
        double s = 0.0;
        for (int b = 0; b < nsum; ++b) s += partial[(((size_t)n * nblocks + b) * 3 + k) * C + c];
        isc[i] = (float)s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float* I = isc; const float* S = isc + NC; const float* T = isc + 2 * NC;
    const int c0 = no_bg ? 1 : 0;
    // weights (loss.py:452-468); reuse coef[0] as the weight scratch
    float wmax = -INFINITY;
    for (int n = 0; n < N; ++n) {
        float rowmax = -INFINITY;
        for (int c = c0; c < C; ++c) {
            float w;
            if (weight_type == 1) w = 1.f / (powf(T[n * C + c], (float)(1.0 / 3.0)) + eps);
            else if (weight_type == 2) w = 1.f / (T[n * C + c] + eps);
            else w = 1.f;
            coef[n * C + c] = w;
            if (weight_type == 2) { const float t = isinf(w) ? 1.f : w; rowmax = fmaxf(rowmax, t); }
        }
        for (int c = c0; c < C; ++c) {
            float w = coef[n * C + c];
            if (weight_type == 2 && isinf(w)) { w = rowmax; coef[n * C + c] = w; }
            wmax = fmaxf(wmax, w);
        }
    }
    float wsum = 0.f, acc = 0.f;
    for (int n = 0; n < N; ++n)
        for (int c = c0; c < C; ++c) {
            const float w = coef[n * C + c] / wmax;
            coef[n * C + c] = w;
            const float score = (2.f * I[n * C + c] + eps) / ((S[n * C + c] + T[n * C + c]) + 2.f * eps);
            acc += w * score; wsum += w;
        }
    loss[0] = 1.f - acc / wsum;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            if (c < c0) { coef[n * C + c] = 0.f; coef[NC + n * C + c] = 0.f; continue; }
            const float w = coef[n * C + c] / wsum;
            const float den = (S[n * C + c] + T[n * C + c]) + 2.f * eps;
            coef[n * C + c] = -w * 2.f / den;
            coef[NC + n * C + c] = w * (2.f * I[n * C + c] + eps) / (den * den);
        }
}

__global__ void dice_bwd_vec_kernel(const float* __restrict__ src, const void* __restrict__ labels, int label_bytes,
                                    const float* __restrict__ soft, const float* __restrict__ coef,
                                    const float* __restrict__ dloss, float* __restrict__ d_src,
                                    int N, long long V, int C, int lpv, int softmax) {
    const long long total = (long long)N * V * lpv;
    const float gl = dloss[0];
    const int NC = N * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % lpv);
        const long long row = i / lpv;
        const int n = (int)(row / V);
        const float4 a = *reinterpret_cast<const float4*>(src + row * C + q * 4);
        float p[4] = {a.x, a.y, a.z, a.w};
        if (softmax) group_softmax4(p, lpv);
        float t[4];
        if (soft) {
            const float4 b = *reinterpret_cast<const float4*>(soft + row * C + q * 4);
            t[0] = b.x; t[1] = b.y; t[2] = b.z; t[3] = b.w;
        } else {
            const int lab = (int)load_label(labels, label_bytes, row) - q * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = (lab == j) ? 1.f : 0.f;
        }
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = coef[n * C + q * 4 + j] * t[j] + coef[NC + n * C + q * 4 + j];
        float4 o;
        if (softmax) {
            float dot = g[0] * p[0] + g[1] * p[1] + g[2] * p[2] + g[3] * p[3];
            for (int k = 1; k < lpv; k <<= 1) dot += __shfl_xor(dot, k);
            o = make_float4(gl * p[0] * (g[0] - dot), gl * p[1] * (g[1] - dot), gl * p[2] * (g[2] - dot), gl * p[3] * (g[3] - dot));
        } else {
            o = make_float4(gl * g[0], gl * g[1], gl * g[2], gl * g[3]);
        }
        *reinterpret_cast<float4*>(d_src + row * C + q * 4) = o;
    }
}

__global__ void dice_bwd_gen_kernel(const float* __restrict__ src, const void* __restrict__ labels, int label_bytes,
                                    const float* __restrict__ soft, const float* __restrict__ coef,
                                    const float* __restrict__ dloss, float* __restrict__ d_src,
                                    int N, long long V, int C, int softmax) {
    const long long total = (long long)N * V;
    const float gl = dloss[0];
    const int NC = N * C;
    for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < total; row += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(row / V);
        const int lab = soft ? -1 : (int)load_label(labels, label_bytes, row);
        float m = -INFINITY, ssum = 1.f, dot = 0.f;
        if (softmax) {
            for (int c = 0; c < C; ++c) m = fmaxf(m, src[row * C + c]);
            ssum = 0.f;
            for (int c = 0; c < C; ++c) ssum += expf(src[row * C + c] - m);
            for (int c = 0; c < C; ++c) {
                const float p = expf(src[row * C + c] - m) / ssum;
                const float t = soft ? soft[row * C + c] : (lab == c ? 1.f : 0.f);
                dot += (coef[n * C + c] * t + coef[NC + n * C + c]) * p;
            }
        }
        for (int c = 0; c < C; ++c) {
            const float t = soft ? soft[row * C + c] : (lab == c ? 1.f : 0.f);
            const float g = coef[n * C + c] * t + coef[NC + n * C + c];
            if (softmax) { const float p = expf(src[row * C + c] - m) / ssum; d_src[row * C + c] = gl * p * (g - dot); }
            else d_src[row * C + c] = gl * g;
        }
    }
}

// standalone softmax over channels and its backward (joint step)
__global__ void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long M, int C, int lpv) {
    if (lpv > 0) {
        const long long total = M * lpv;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const int q = (int)(i % lpv); const long long row = i / lpv;
            const float4 a = *reinterpret_cast<const float4*>(x + row * C + q * 4);
            float p[4] = {a.x, a.y, a.z, a.w};
            group_softmax4(p, lpv);
            *reinterpret_cast<float4*>(y + row * C + q * 4) = make_float4(p[0], p[1], p[2], p[3]);
        }
    } else {
        for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < M; row += (long long)gridDim.x * blockDim.x) {
            float m = -INFINITY, s = 0.f;
            for (int c = 0; c < C; ++c) m = fmaxf(m, x[row * C + c]);
            for (int c = 0; c < C; ++c) s += expf(x[row * C + c] - m);
            for (int c = 0; c < C; ++c) y[row * C + c] = expf(x[row * C + c] - m) / s;
        }
    }
}

__global__ void softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                   long long M, int C, int lpv) {
    if (lpv > 0) {
        const long long total = M * lpv;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const int q = (int)(i % lpv); const long long row = i / lpv;
            const float4 g = *reinterpret_cast<const float4*>(dy + row * C + q * 4);
            const float4 p = *reinterpret_cast<const float4*>(y + row * C + q * 4);
            float dot = g.x * p.x + g.y * p.y + g.z * p.z + g.w * p.w;
            for (int k = 1; k < lpv; k <<= 1) dot += __shfl_xor(dot, k);
            *reinterpret_cast<float4*>(dx + row * C + q * 4) = make_float4(p.x * (g.x - dot), p.y * (g.y - dot), p.z * (g.z - dot), p.w * (g.w - dot));
        }
    } else {
        for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < M; row += (long long)gridDim.x * blockDim.x) {
            float dot = 0.f;
            for (int c = 0; c < C; ++c) dot += dy[row * C + c] * y[row * C + c];
            for (int c = 0; c < C; ++c) dx[row * C + c] = y[row * C + c] * (dy[row * C + c] - dot);
        }
    }
}

__global__ void one_hot_kernel(const void* __restrict__ labels, int label_bytes, float* __restrict__ out, long long M, int C) {
    const long long total = M * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / C; const int c = (int)(i % C);
        out[i] = ((int)load_label(labels, label_bytes, row) == c) ? 1.f : 0.f;
    }
}

static int lpv_for(int C) {
    if (C % 4 != 0) return 0;
    const int q = C / 4;
    if (q > 64 || (q & (q - 1)) != 0) return 0;
    return q;
}

// ------------------------------------------------------------------------------------------------
// NCC
// ------------------------------------------------------------------------------------------------
__global__ void ncc_partial_kernel(const float* __restrict__ x, const float* __restrict__ y, long long V,
                                   double* __restrict__ partial /* [N][blocks][5] */) {
    __shared__ double red[5][4];
    const int n = blockIdx.y;
    const float* xs = x + (long long)n * V; const float* ys = y + (long long)n * V;
    float a[5] = {0, 0, 0, 0, 0};
    double acc[5] = {0, 0, 0, 0, 0};
    const long long V4 = V / 4;
    int cnt = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < V4; i += (long long)gridDim.x * blockDim.x) {
        const float4 p = reinterpret_cast<const float4*>(xs)[i];
        const float4 q = reinterpret_cast<const float4*>(ys)[i];
        a[0] += (p.x + p.y) + (p.z + p.w);
        a[1] += (q.x + q.y) + (q.z + q.w);
        a[2] += (p.x * q.x + p.y * q.y) + (p.z * q.z + p.w * q.w);
        a[3] += (p.x * p.x + p.y * p.y) + (p.z * p.z + p.w * p.w);
        a[4] += (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
        if (++cnt == 16) {   // flush fp32 partials into double every 64 elements
#pragma unroll
            for (int k = 0; k < 5; ++k) { acc[k] += (double)a[k]; a[k] = 0.f; }
            cnt = 0;
        }
    }
    if (blockIdx.x == 0) for (long long i = V4 * 4 + threadIdx.x; i < V; i += blockDim.x) {
        const float p = xs[i], q = ys[i];
        a[0] += p; a[1] += q; a[2] += p * q; a[3] += p * p; a[4] += q * q;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] += (double)a[k];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) { const double w = da_wave_sum(acc[k]); if (lane == 0) red[k][wid] = w; }
    __syncthreads();
    if (threadIdx.x < 5) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[threadIdx.x][w];
        partial[((size_t)n * gridDim.x + blockIdx.x) * 5 + threadIdx.x] = s;
    }
}

// stats[n] = {mean_x, mean_y, cov, var_x, var_y, ncc, M, 0}
__global__ void ncc_finalize_kernel(const double* __restrict__ partial, int nblocks, int nsum, int N, long long V,
                                    float* __restrict__ loss, double* __restrict__ stats) {
    __shared__ double sncc[64];
    const int n = threadIdx.x;
    if (n < N) {
        double s[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < nsum; ++b)
            for (int k = 0; k < 5; ++k) s[k] += partial[((size_t)n * nblocks + b) * 5 + k];
        const double M = (double)V;
        const double mx = s[0] / M, my = s[1] / M;
        const double cov = s[2] / M - mx * my;
        double vx = s[3] / M - mx * mx, vy = s[4] / M - my * my;
        const double ncc = cov / (sqrt(vx) * sqrt(vy));
        double* o = stats + (size_t)n * 8;
        o[0] = mx; o[1] = my; o[2] = cov; o[3] = vx; o[4] = vy; o[5] = ncc; o[6] = M; o[7] = 0.0;
        sncc[n] = ncc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < N; ++i) t += sncc[i];
        loss[0] = (float)(1.0 - t / (double)N);
    }
}

__global__ void ncc_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const double* __restrict__ stats,
                               const float* __restrict__ dloss, float* __restrict__ dx, float* __restrict__ dy,
                               int N, long long V) {
    const int n = blockIdx.y;
    const double* s = stats + (size_t)n * 8;
    const float mx = (float)s[0], my = (float)s[1];
    const double sxy = sqrt(s[3]) * sqrt(s[4]);
    const double k = -(double)dloss[0] / ((double)N * s[6]);
    // d ncc / dx_i = (1/M) [ ytil_i / (sx sy) - ncc * xtil_i / vx ]
    const float ax = (float)(k / sxy), bx = (float)(-k * s[5] / s[3]);
    const float ay = (float)(k / sxy), by = (float)(-k * s[5] / s[4]);
    const float* xs = x + (long long)n * V; const float* ys = y + (long long)n * V;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long long)gridDim.x * blockDim.x) {
        const float xt = xs[i] - mx, yt = ys[i] - my;
        if (dx) dx[(long long)n * V + i] = ax * yt + bx * xt;
        if (dy) dy[(long long)n * V + i] = ay * xt + by * yt;
    }
}

// ------------------------------------------------------------------------------------------------
// Bending energy.  K[c][term] folds the reference's channel-broadcast weights and all the means.
// terms: 0 dd(D) 1 dd(H) 2 dd(W) 3 mixed(D,H) 4 mixed(H,W) 5 mixed(D,W)
// ------------------------------------------------------------------------------------------------
struct BendK { float k[3][6]; };

static BendK bending_coeffs(int N, int D, int H, int W, const float* spacing3, int normalize, int norm) {
    float sp[3] = {1.f, 1.f, 1.f};
    if (spacing3) { sp[0] = spacing3[0]; sp[1] = spacing3[1]; sp[2] = spacing3[2]; }
    if (normalize) { const float m = fminf(sp[0], fminf(sp[1], sp[2])); sp[0] /= m; sp[1] /= m; sp[2] /= m; }
    float dims[3] = {(float)D, (float)H, (float)W};
    if (normalize) { const float m = fminf(dims[0], fminf(dims[1], dims[2])); dims[0] /= m; dims[1] /= m; dims[2] /= m; }
    const float den[6] = {sp[0] * sp[0], sp[1] * sp[1], sp[2] * sp[2], sp[0] * sp[1], sp[1] * sp[2], sp[2] * sp[0]};
    const double Mint = (double)(D - 2) * (H - 2) * (W - 2);
    BendK K;
    for (int c = 0; c < 3; ++c)
        for (int t = 0; t < 6; ++t) {
            // loss.py:722-727 (vector over the channel axis); any norm other than 'L2' skips that block: plain means of the |differences| (:729)
            const float w = norm == 2 ? dims[c] * sp[c] / den[t] : 1.f;
            const double mult = (t < 3 ? 1.0 : 2.0) / (9.0 * 3.0 * (double)N * Mint);   // .mean(2), .mean(), /9, 2x mixed
            K.k[c][t] = (float)((double)(w * w) * mult);
        }
    return K;
}

#define BU(dd, hh, ww) u[((((long long)(dd)) * H + (hh)) * W + (ww)) * 3 + c]

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 ld3(const float* __restrict__ u, long long vox) {          // one voxel's three displacement components (12 contiguous bytes)
    const float* p = u + vox * 3;
    return F3{p[0], p[1], p[2]};
}

// One thread per interior voxel, all three channels: the 19 points of the six difference stencils are 12-byte voxel reads that
// neighbouring threads share cache lines on (the per-element form read every point as a stride-3 scalar).
template <bool L1>
__global__ void bending_partial_kernel(const float* __restrict__ disp, int D, int H, int W, BendK K,
                                       double* __restrict__ partial) {
    __shared__ double red[4];
    const int n = blockIdx.y;
    const float* u = disp + (long long)n * D * H * W * 3;
    const int Di = D - 2, Hi = H - 2, Wi = W - 2;
    const long long total = (long long)Di * Hi * Wi;
    const long long sH = W, sD = (long long)H * W;
    float acc = 0.f; double dacc = 0.0; int cnt = 0;
    for (DaXcdLoop L = da_xcd_loop(total); L.i < L.end; L.i += L.step) {      // (z-neighbour planes stay inside one XCD's L2)
        const long long i = L.i;
        long long r = i;
        const int w = (int)(r % Wi) + 1; r /= Wi;
        const int h = (int)(r % Hi) + 1; const int d = (int)(r / Hi) + 1;
        const long long p = ((long long)d * H + h) * W + w;
        const F3 c0 = ld3(u, p);
        const F3 dp = ld3(u, p + sD), dm = ld3(u, p - sD), hp = ld3(u, p + sH), hm = ld3(u, p - sH), wp = ld3(u, p + 1), wm = ld3(u, p - 1);
        const F3 dhpp = ld3(u, p + sD + sH), dhmm = ld3(u, p - sD - sH), dhpm = ld3(u, p + sD - sH), dhmp = ld3(u, p - sD + sH);
        const F3 hwpp = ld3(u, p + sH + 1), hwmm = ld3(u, p - sH - 1), hwpm = ld3(u, p + sH - 1), hwmp = ld3(u, p - sH + 1);
        const F3 dwpp = ld3(u, p + sD + 1), dwmm = ld3(u, p - sD - 1), dwpm = ld3(u, p + sD - 1), dwmp = ld3(u, p - sD + 1);
#define BEND_CH(f, c)                                                                                   \
        {                                                                                               \
            const float t0 = dp.f + dm.f - 2.f * c0.f, t1 = hp.f + hm.f - 2.f * c0.f, t2 = wp.f + wm.f - 2.f * c0.f;   \
            const float t3 = dhpp.f + dhmm.f - dhpm.f - dhmp.f;                                         \
            const float t4 = hwpp.f + hwmm.f - hwpm.f - hwmp.f;                                         \
            const float t5 = dwpp.f + dwmm.f - dwpm.f - dwmp.f;                                         \
            if (L1) acc += K.k[c][0] * fabsf(t0) + K.k[c][1] * fabsf(t1) + K.k[c][2] * fabsf(t2) + K.k[c][3] * fabsf(t3) + K.k[c][4] * fabsf(t4) + K.k[c][5] * fabsf(t5); \
            else acc += K.k[c][0] * t0 * t0 + K.k[c][1] * t1 * t1 + K.k[c][2] * t2 * t2 + K.k[c][3] * t3 * t3 + K.k[c][4] * t4 * t4 + K.k[c][5] * t5 * t5; \
        }
        BEND_CH(x, 0) BEND_CH(y, 1) BEND_CH(z, 2)
#undef BEND_CH
        if (++cnt == 16) { dacc += (double)acc; acc = 0.f; cnt = 0; }
    }
    dacc += (double)acc;
    const double s = da_block_sum(dacc, red);
    if (threadIdx.x == 0) partial[(size_t)n * gridDim.x + blockIdx.x] = s;
}

__global__ void scalar_finalize_kernel(const double* __restrict__ partial, int count, float* __restrict__ loss) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += blockDim.x) s += partial[i];
    s = da_block_sum(s, red);
    if (threadIdx.x == 0) loss[0] = (float)s;
}

// gather-form gradient of one element: d loss / d u[p] = 2 * sum_terms K * sum_{centres q containing p} term(q) * coef  (L1: sign(term(q)), no 2)
template <bool L1>
__device__ __forceinline__ float bend_grad_elem(const float* __restrict__ u, int d, int h, int w, int c, int D, int H, int W, const BendK& K) {
    auto interior = [&](int dd, int hh, int ww) { return dd >= 1 && dd < D - 1 && hh >= 1 && hh < H - 1 && ww >= 1 && ww < W - 1; };
    auto f = [](float t) { return L1 ? (t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f)) : t; };      // d|t|/dt = sign(t) (0 at 0, torch.abs)
    float g = 0.f;
    {   // second differences: centres p-e (+1), p+e (+1), p (-2) along each axis
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (interior(d, h, w)) {
            const float u0 = BU(d, h, w);
            s0 -= 2.f * f(BU(d + 1, h, w) + BU(d - 1, h, w) - 2.f * u0);
            s1 -= 2.f * f(BU(d, h + 1, w) + BU(d, h - 1, w) - 2.f * u0);
            s2 -= 2.f * f(BU(d, h, w + 1) + BU(d, h, w - 1) - 2.f * u0);
        }
#pragma unroll
        for (int sgn = -1; sgn <= 1; sgn += 2) {
            if (interior(d + sgn, h, w)) s0 += f(BU(d + 2 * sgn, h, w) + BU(d, h, w) - 2.f * BU(d + sgn, h, w));
            if (interior(d, h + sgn, w)) s1 += f(BU(d, h + 2 * sgn, w) + BU(d, h, w) - 2.f * BU(d, h + sgn, w));
            if (interior(d, h, w + sgn)) s2 += f(BU(d, h, w + 2 * sgn) + BU(d, h, w) - 2.f * BU(d, h, w + sgn));
        }
        g += K.k[c][0] * s0 + K.k[c][1] * s1 + K.k[c][2] * s2;
    }
    {   // mixed differences: centre q = p - (sa*ea + sb*eb) has coefficient sa*sb for u[p]
        float s3 = 0.f, s4 = 0.f, s5 = 0.f;
#pragma unroll
        for (int sa = -1; sa <= 1; sa += 2)
#pragma unroll
            for (int sb = -1; sb <= 1; sb += 2) {
                const float cf = (float)(sa * sb);
                { const int qd = d - sa, qh = h - sb;   // (D,H)
                  if (interior(qd, qh, w)) s3 += cf * f(BU(qd + 1, qh + 1, w) + BU(qd - 1, qh - 1, w) - BU(qd + 1, qh - 1, w) - BU(qd - 1, qh + 1, w)); }
                { const int qh = h - sa, qw = w - sb;   // (H,W)
                  if (interior(d, qh, qw)) s4 += cf * f(BU(d, qh + 1, qw + 1) + BU(d, qh - 1, qw - 1) - BU(d, qh + 1, qw - 1) - BU(d, qh - 1, qw + 1)); }
                { const int qd = d - sa, qw = w - sb;   // (D,W)
                  if (interior(qd, h, qw)) s5 += cf * f(BU(qd + 1, h, qw + 1) + BU(qd - 1, h, qw - 1) - BU(qd + 1, h, qw - 1) - BU(qd - 1, h, qw + 1)); }
            }
        g += K.k[c][3] * s3 + K.k[c][4] * s4 + K.k[c][5] * s5;
    }
    return g;
}

// One thread per element (c fastest: a wave instruction reads 256 contiguous bytes).  'L2', voxel at least two away from every face
// (all 25 stencil centres that contain it are interior): the composition S^T S of each difference stencil with its own adjoint is a
// FIXED stencil --
//   second difference along e:  u(p+2e) - 4 u(p+e) + 6 u(p) - 4 u(p-e) + u(p-2e)
//   mixed difference in (a, b): 4 u(p) - 2 [u(p+-2a) + u(p+-2b)] + [u(p+2a+2b) + u(p+2a-2b) + u(p-2a+2b) + u(p-2a-2b)]
// -- 25 reads instead of ~110 for the term-by-term gather.  The shell within two voxels of a face (and 'L1', whose sign() does not
// commute with the composition) takes the term-by-term form.
template <bool L1>
__global__ void bending_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ dloss,
                                   float* __restrict__ d_disp, int D, int H, int W, BendK K) {
    const int n = blockIdx.y;
    const float* u = disp + (long long)n * D * H * W * 3;
    float* du = d_disp + (long long)n * D * H * W * 3;
    const long long total = (long long)D * H * W * 3;
    const float gl = (L1 ? 1.f : 2.f) * dloss[0];
    const long long sW = 3, sH = (long long)W * 3, sD = (long long)H * W * 3;
    for (DaXcdLoop L = da_xcd_loop(total, 768); L.i < L.end; L.i += L.step) {
        const long long i = L.i;
        const int c = (int)(i % 3); long long r = i / 3;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const int d = (int)(r / H);
        float g;
        const bool deep = !L1 && d >= 2 && d < D - 2 && h >= 2 && h < H - 2 && w >= 2 && w < W - 2;
        if (deep) {
            const float* q = u + i;
            const float c0 = q[0];
            const float d2 = q[2 * sD] + q[-2 * sD], d1 = q[sD] + q[-sD];
            const float h2 = q[2 * sH] + q[-2 * sH], h1 = q[sH] + q[-sH];
            const float w2 = q[2 * sW] + q[-2 * sW], w1 = q[sW] + q[-sW];
            const float dh = q[2 * sD + 2 * sH] + q[2 * sD - 2 * sH] + q[-2 * sD + 2 * sH] + q[-2 * sD - 2 * sH];
            const float hw = q[2 * sH + 2 * sW] + q[2 * sH - 2 * sW] + q[-2 * sH + 2 * sW] + q[-2 * sH - 2 * sW];
            const float dw = q[2 * sD + 2 * sW] + q[2 * sD - 2 * sW] + q[-2 * sD + 2 * sW] + q[-2 * sD - 2 * sW];
            g = K.k[c][0] * (d2 - 4.f * d1 + 6.f * c0) + K.k[c][1] * (h2 - 4.f * h1 + 6.f * c0) + K.k[c][2] * (w2 - 4.f * w1 + 6.f * c0) +
                K.k[c][3] * (4.f * c0 - 2.f * (d2 + h2) + dh) + K.k[c][4] * (4.f * c0 - 2.f * (h2 + w2) + hw) + K.k[c][5] * (4.f * c0 - 2.f * (d2 + w2) + dw);
        } else {
            g = bend_grad_elem<L1>(u, d, h, w, c, D, H, W, K);
        }
        du[i] = gl * g;
    }
}
#undef BU

// ------------------------------------------------------------------------------------------------
// eval: argmax (first maximum index, torch.max) + integer overlap counts per class
// ------------------------------------------------------------------------------------------------
__global__ void argmax_counts_kernel(const float* __restrict__ logits, const void* __restrict__ truth, int label_bytes,
                                     long long V, int C, int lpv, unsigned long long* __restrict__ counts,
                                     unsigned char* __restrict__ pred) {
    extern __shared__ unsigned int shc[];   // [3][C]
    const int n = blockIdx.y;
    for (int c = threadIdx.x; c < 3 * C; c += blockDim.x) shc[c] = 0u;
    __syncthreads();
    const int L = lpv > 0 ? lpv : 1;
    const long long total = V * L;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % L); const long long v = i / L;
        const long long row = (long long)n * V + v;
        float best; int bi;
        if (lpv > 0) {
            const float4 a = *reinterpret_cast<const float4*>(logits + row * C + q * 4);
            best = a.x; bi = q * 4;
            if (a.y > best) { best = a.y; bi = q * 4 + 1; }
            if (a.z > best) { best = a.z; bi = q * 4 + 2; }
            if (a.w > best) { best = a.w; bi = q * 4 + 3; }
            for (int o = 1; o < lpv; o <<= 1) {
                const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
        } else {
            best = logits[row * C]; bi = 0;
            for (int c = 1; c < C; ++c) { const float a = logits[row * C + c]; if (a > best) { best = a; bi = c; } }
        }
        if (q == 0) {
            const int t = (int)load_label(truth, label_bytes, row);
            atomicAdd(&shc[bi], 1u);
            if (t >= 0 && t < C) atomicAdd(&shc[C + t], 1u);
            if (t == bi) atomicAdd(&shc[2 * C + bi], 1u);
            if (pred) pred[row] = (unsigned char)bi;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * C; c += blockDim.x) {
        const int k = c / C, cc = c % C;
        if (shc[c]) atomicAdd(&counts[((size_t)n * C + cc) * 3 + k], (unsigned long long)shc[c]);
    }
}


// Overlap counts of two LABEL maps (lib/evalMetrics.py:103-217, lib/loss.py:348-391 -- every metric there
// is a function of |P==c|, |T==c|, |P==c & T==c|).  Each thread walks RUN consecutive voxels and merges runs of equal
// (pred, truth) pairs in registers before touching the per-wave LDS histograms: anatomical label maps are piecewise constant,
// so this removes most of the same-address LDS atomic serialisation.  Exact integer arithmetic; 64-bit global atomics.
__global__ void label_overlap_counts_kernel(const void* __restrict__ pred, int pred_bytes, const void* __restrict__ truth, int truth_bytes,
                                            long long V, int C, unsigned long long* __restrict__ counts) {
    extern __shared__ unsigned int shc[];   // [4 waves][3][C]
    constexpr int RUN = 16;
    const int n = blockIdx.y;
    for (int c = threadIdx.x; c < 4 * 3 * C; c += blockDim.x) shc[c] = 0u;
    __syncthreads();
    unsigned int* h = shc + (threadIdx.x >> 6) * 3 * C;
    auto flush = [&](int pl, int tl, unsigned int len) {
        if (len == 0u) return;
        if (pl >= 0 && pl < C) atomicAdd(&h[pl], len);
        if (tl >= 0 && tl < C) atomicAdd(&h[C + tl], len);
        if (pl == tl && pl >= 0 && pl < C) atomicAdd(&h[2 * C + pl], len);
    };
    const long long nruns = (V + RUN - 1) / RUN;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < nruns; r += (long long)gridDim.x * blockDim.x) {
        const long long v0 = r * RUN;
        const int cnt = (int)((V - v0) < RUN ? (V - v0) : RUN);
        int pl = -1, tl = -1; unsigned int len = 0u;
        for (int k = 0; k < cnt; ++k) {
            const long long row = (long long)n * V + v0 + k;
            const int a = (int)load_label(pred, pred_bytes, row), b = (int)load_label(truth, truth_bytes, row);
            if (a != pl || b != tl) { flush(pl, tl, len); pl = a; tl = b; len = 0u; }
            ++len;
        }
        flush(pl, tl, len);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * C; c += blockDim.x) {
        const unsigned int t = shc[c] + shc[3 * C + c] + shc[6 * C + c] + shc[9 * C + c];
        const int k = c / C, cc = c % C;
        if (t) atomicAdd(&counts[((size_t)n * C + cc) * 3 + k], (unsigned long long)t);
    }
}

}  // namespace

// ================================================================================================
extern "C" size_t da_dice_ws_bytes(int N, long long V, int C) {
    (void)V;
    return da_align((size_t)N * kDiceBlocks * 3 * C * sizeof(double)) + da_align((size_t)3 * N * C * sizeof(float));
}

extern "C" int da_dice_fwd(const float* src, const void* labels, int label_bytes, const float* soft_target,
                           int N, long long V, int C, int softmax, int weight_type, int no_bg, float eps,
                           float* loss, float* coef, void* ws, size_t ws_bytes, void* stream) {
    if (!src || (!labels && !soft_target) || !loss || !coef || N <= 0 || N > 64 || V <= 0 || C <= 0 || C > 256) return DA_ERR_BADARG;
    if (labels && label_bytes != 1 && label_bytes != 8) return DA_ERR_BADARG;
    if (ws_bytes < da_dice_ws_bytes(N, V, C)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    double* partial = (double*)ws;
    float* isc = (float*)((char*)ws + da_align((size_t)N * kDiceBlocks * 3 * C * sizeof(double)));
    const int lpv = lpv_for(C);
    int nblocks = (int)da_cdiv(V, 256); if (nblocks > kDiceBlocks) nblocks = kDiceBlocks;
    if (lpv > 0) {
        const int slots = 256 / lpv;
        hipLaunchKernelGGL(dice_partial_vec_kernel, dim3(nblocks, N), dim3(256), (size_t)3 * slots * C * sizeof(float), st,
                           src, labels, label_bytes, soft_target, V, C, lpv, softmax, partial);
    } else {
        hipLaunchKernelGGL(dice_partial_gen_kernel, dim3(nblocks, N), dim3(256), (size_t)4 * 3 * C * sizeof(float), st,
                           src, labels, label_bytes, soft_target, V, C, softmax, partial);
    }
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_inplace_kernel, dim3((unsigned)da_cdiv((long long)N * 3 * C * 64, 256)), dim3(256), 0, st, partial, N, nblocks, 3 * C);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nblocks, 1, N, C, weight_type, no_bg, eps, loss, coef, isc);
    DA_LAUNCH_CHECK();
    return 0;
}

// Dice(softmax(src), labels) exactly as da_dice_fwd(softmax = 1) AND prob = softmax(src) written out, in one pass over src.
extern "C" int da_softmax_dice_fwd(const float* src, const void* labels, int label_bytes, float* prob,
                                   int N, long long V, int C, int weight_type, int no_bg, float eps,
                                   float* loss, float* coef, void* ws, size_t ws_bytes, void* stream) {
    if (!src || !labels || !prob || !loss || !coef || N <= 0 || N > 64 || V <= 0 || C <= 0 || C > 256) return DA_ERR_BADARG;
    if (label_bytes != 1 && label_bytes != 8) return DA_ERR_BADARG;
    const int lpv = lpv_for(C);
    if (lpv <= 0) return DA_ERR_UNSUPPORTED;                               // callers run da_dice_fwd + da_softmax_fwd
    if (ws_bytes < da_dice_ws_bytes(N, V, C)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    double* partial = (double*)ws;
    float* isc = (float*)((char*)ws + da_align((size_t)N * kDiceBlocks * 3 * C * sizeof(double)));
    int nblocks = (int)da_cdiv(V, 256); if (nblocks > kDiceBlocks) nblocks = kDiceBlocks;
    const int slots = 256 / lpv;
    hipLaunchKernelGGL(dice_partial_vec_kernel, dim3(nblocks, N), dim3(256), (size_t)3 * slots * C * sizeof(float), st,
                       src, labels, label_bytes, (const float*)nullptr, V, C, lpv, 1, partial, prob);
    DA_LAUNCH_CHECK();
    return da_dice_finish(partial, nblocks, N, C, weight_type, no_bg, eps, loss, coef, isc, st);
}

// Dice from per-block partial sums [N][nblocks][3][C] (I, S, T) produced by another kernel (the fused label-warp Dice, warp.hip):
// reduce over the blocks, then weights / loss / backward coefficients exactly as da_dice_fwd.  isc: 3*N*C floats of scratch.
int da_dice_finish(double* partial, int nblocks, int N, int C, int weight_type, int no_bg, float eps,
                   float* loss, float* coef, float* isc, hipStream_t st) {
    hipLaunchKernelGGL(colreduce_inplace_kernel, dim3((unsigned)da_cdiv((long long)N * 3 * C * 64, 256)), dim3(256), 0, st, partial, N, nblocks, 3 * C);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nblocks, 1, N, C, weight_type, no_bg, eps, loss, coef, isc);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_dice_bwd(const float* src, const void* labels, int label_bytes, const float* soft_target,
                           const float* coef, const float* dloss, float* d_src,
                           int N, long long V, int C, int softmax, void* stream) {
    if (!src || (!labels && !soft_target) || !coef || !dloss || !d_src || N <= 0 || V <= 0 || C <= 0) return DA_ERR_BADARG;
    hipStream_t st = da_stream(stream);
    const int lpv = lpv_for(C);
    if (lpv > 0) {
        const long long total = (long long)N * V * lpv;
        hipLaunchKernelGGL(dice_bwd_vec_kernel, dim3(da_grid(total, 256)), dim3(256), 0, st, src, labels, label_bytes, soft_target, coef, dloss, d_src, N, V, C, lpv, softmax);
    } else {
        hipLaunchKernelGGL(dice_bwd_gen_kernel, dim3(da_grid((long long)N * V, 256)), dim3(256), 0, st, src, labels, label_bytes, soft_target, coef, dloss, d_src, N, V, C, softmax);
    }
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_softmax_fwd(const float* x, float* y, long long M, int C, void* stream) {
    if (!x || !y || M <= 0 || C <= 0) return DA_ERR_BADARG;
    const int lpv = lpv_for(C);
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(da_grid(M * (lpv > 0 ? lpv : 1), 256)), dim3(256), 0, da_stream(stream), x, y, M, C, lpv);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_softmax_bwd(const float* dy, const float* y, float* dx, long long M, int C, void* stream) {
    if (!dy || !y || !dx || M <= 0 || C <= 0) return DA_ERR_BADARG;
    const int lpv = lpv_for(C);
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(da_grid(M * (lpv > 0 ? lpv : 1), 256)), dim3(256), 0, da_stream(stream), dy, y, dx, M, C, lpv);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_one_hot(const void* labels, int label_bytes, float* out, long long M, int C, void* stream) {
    if (!labels || !out || M <= 0 || C <= 0 || (label_bytes != 1 && label_bytes != 8)) return DA_ERR_BADARG;
    hipLaunchKernelGGL(one_hot_kernel, dim3(da_grid(M * C, 256)), dim3(256), 0, da_stream(stream), labels, label_bytes, out, M, C);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t da_ncc_ws_bytes(int N, long long V) {
    (void)V;
    return da_align((size_t)N * kBlocks * 5 * sizeof(double));
}

extern "C" int da_ncc_fwd(const float* x, const float* y, int N, long long V, float* loss, double* stats,
                          void* ws, size_t ws_bytes, void* stream) {
    if (!x || !y || !loss || !stats || N <= 0 || N > 64 || V <= 0) return DA_ERR_BADARG;
    if (ws_bytes < da_ncc_ws_bytes(N, V)) return DA_ERR_WS_SMALL;
    if ((V % 4) != 0 && N > 1) return DA_ERR_UNSUPPORTED;   // per-sample base must stay 16-byte aligned
    hipStream_t st = da_stream(stream);
    int nblocks = (int)da_cdiv(V / 4 + 1, 256 * 4); if (nblocks > kBlocks) nblocks = kBlocks; if (nblocks < 1) nblocks = 1;
    hipLaunchKernelGGL(ncc_partial_kernel, dim3(nblocks, N), dim3(256), 0, st, x, y, V, (double*)ws);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_inplace_kernel, dim3((unsigned)da_cdiv((long long)N * 5 * 64, 256)), dim3(256), 0, st, (double*)ws, N, nblocks, 5);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(ncc_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, nblocks, 1, N, V, loss, stats);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_ncc_bwd(const float* x, const float* y, const double* stats, const float* dloss,
                          float* dx, float* dy, int N, long long V, void* stream) {
    if (!x || !y || !stats || !dloss || N <= 0 || V <= 0) return DA_ERR_BADARG;
    if (!dx && !dy) return 0;
    hipLaunchKernelGGL(ncc_bwd_kernel, dim3(da_grid(V, 256, 2048), N), dim3(256), 0, da_stream(stream), x, y, stats, dloss, dx, dy, N, V);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t da_bending_ws_bytes(int N, int D, int H, int W) {
    (void)D; (void)H; (void)W;
    return da_align((size_t)N * kBendBlocks * sizeof(double));
}

extern "C" int da_bending_fwd(const float* disp, int N, int D, int H, int W, const float* spacing3, int normalize, int norm,
                              float* loss, void* ws, size_t ws_bytes, void* stream) {
    if (!disp || !loss || N <= 0 || D < 3 || H < 3 || W < 3 || (norm != 1 && norm != 2)) return DA_ERR_BADARG;
    if (ws_bytes < da_bending_ws_bytes(N, D, H, W)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    const BendK K = bending_coeffs(N, D, H, W, spacing3, normalize, norm);
    const long long total = (long long)(D - 2) * (H - 2) * (W - 2);
    int nblocks = (int)da_cdiv(total, 256); if (nblocks > kBendBlocks) nblocks = kBendBlocks; if (nblocks < 1) nblocks = 1;
    if (norm == 2) hipLaunchKernelGGL((bending_partial_kernel<false>), dim3(nblocks, N), dim3(256), 0, st, disp, D, H, W, K, (double*)ws);
    else hipLaunchKernelGGL((bending_partial_kernel<true>), dim3(nblocks, N), dim3(256), 0, st, disp, D, H, W, K, (double*)ws);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(scalar_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nblocks * N, loss);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_bending_bwd(const float* disp, const float* dloss, float* d_disp, int N, int D, int H, int W,
                              const float* spacing3, int normalize, int norm, void* stream) {
    if (!disp || !dloss || !d_disp || N <= 0 || D < 3 || H < 3 || W < 3 || (norm != 1 && norm != 2)) return DA_ERR_BADARG;
    const BendK K = bending_coeffs(N, D, H, W, spacing3, normalize, norm);
    const long long total = (long long)D * H * W * 3;
    if (norm == 2) hipLaunchKernelGGL((bending_bwd_kernel<false>), dim3(da_grid(total, 256, 4096), N), dim3(256), 0, da_stream(stream), disp, dloss, d_disp, D, H, W, K);
    else hipLaunchKernelGGL((bending_bwd_kernel<true>), dim3(da_grid(total, 256, 4096), N), dim3(256), 0, da_stream(stream), disp, dloss, d_disp, D, H, W, K);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_argmax_dice_counts(const float* logits, const void* truth, int label_bytes, int N, long long V, int C,
                                     unsigned long long* counts, unsigned char* pred, void* stream) {
    if (!logits || !truth || !counts || N <= 0 || V <= 0 || C <= 0 || C > 256 || (label_bytes != 1 && label_bytes != 8)) return DA_ERR_BADARG;
    const int lpv = lpv_for(C);
    const long long total = V * (lpv > 0 ? lpv : 1);
    hipLaunchKernelGGL(argmax_counts_kernel, dim3(da_grid(total, 256, 1024), N), dim3(256), (size_t)3 * C * sizeof(unsigned int), da_stream(stream),
                       logits, truth, label_bytes, V, C, lpv, counts, pred);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_label_overlap_counts(const void* pred, int pred_bytes, const void* truth, int truth_bytes, int N, long long V, int C,
                                       unsigned long long* counts, void* stream) {
    if (!pred || !truth || !counts || N <= 0 || V <= 0 || C <= 0 || C > 1024 || (pred_bytes != 1 && pred_bytes != 8) || (truth_bytes != 1 && truth_bytes != 8))
        return DA_ERR_BADARG;
    hipLaunchKernelGGL(label_overlap_counts_kernel, dim3(da_grid((V + 15) / 16, 256, 1024), N), dim3(256), (size_t)4 * 3 * C * sizeof(unsigned int),
                       da_stream(stream), pred, pred_bytes, truth, truth_bytes, V, C, counts);
    DA_LAUNCH_CHECK();
    return 0;
}
