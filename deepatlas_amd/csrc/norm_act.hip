// BatchNorm3d (train/eval) + LeakyReLU/ReLU forward & backward, per-channel column sums.
// Rows a1/a3 of SURVEY.md §8: nn.BatchNorm3d + nn.LeakyReLU at unets.py:31-32,51-52.
// Layout: x[M][C] (NDHWC flattened, M = N*D*H*W).  All kernels are HBM-bound streaming passes:
// 16-byte accesses per lane, per-thread fp32 partials, per-block and cross-block sums in double,
// cross-block reduction in a second tiny launch (deterministic, no atomics).
#include "common.h"

namespace {

constexpr int kMaxBlocks = 1024;

struct RowPlan { int vec, cq, rpi, block, grid; long long rows_per_block; };

static RowPlan plan_rows(long long M, int C) {
    RowPlan p;
    p.vec = (C % 4 == 0) ? 4 : 1;
    p.cq = C / p.vec;
    p.rpi = 256 / p.cq; if (p.rpi < 1) p.rpi = 1;
    p.block = p.cq * p.rpi;
    long long g = da_cdiv(M, (long long)p.rpi * 16);
    if (g > kMaxBlocks) g = kMaxBlocks;
    if (g < 1) g = 1;
    p.grid = (int)g;
    p.rows_per_block = da_cdiv(M, g);
    return p;
}

// partial[b][k][C] doubles, k < NK.  MODE 0: (sum x, sum x^2); MODE 1: (sum x); MODE 2: BN backward sums
template <int VEC, int MODE, typename T = float>
__global__ void col_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                   const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                   long long M, int C, long long rows_per_block, double* __restrict__ partial) {
    extern __shared__ double sh[];   // [2][rpi][C]
    const int cq = C / VEC;
    const int rpi = blockDim.x / cq;
    const int q = threadIdx.x % cq, r = threadIdx.x / cq;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    float a0[VEC], a1[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
    float mu[VEC], rs[VEC], sc[VEC], sf[VEC];
    if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { mu[j] = mean[q * VEC + j]; rs[j] = rstd[q * VEC + j]; sc[j] = scale[q * VEC + j]; sf[j] = shift[q * VEC + j]; }
    }
    long long row = r0 + r;
    if (VEC == 4 && MODE == 2) {
        // BatchNorm-backward sums: two rows in flight per lane (unconditional loads, the second row clamped and masked)
        for (; row < r1; row += 2 * rpi) {
            const long long rb = row + rpi < r1 ? row + rpi : row;
            const float live = row + rpi < r1 ? 1.f : 0.f;
            const float4 t0 = da_ldq(x, row * cq + q), g0 = da_ldq(dy, row * cq + q), t1 = da_ldq(x, rb * cq + q), g1 = da_ldq(dy, rb * cq + q);
            const float xa[2][4] = {{t0.x, t0.y, t0.z, t0.w}, {t1.x, t1.y, t1.z, t1.w}}, ga[2][4] = {{g0.x, g0.y, g0.z, g0.w}, {g1.x * live, g1.y * live, g1.z * live, g1.w * live}};
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float z = xa[u][j] * sc[j] + sf[j];
                    const float dz = ga[u][j] * da_act_grad(z, slope);
                    a0[j] += dz; a1[j] += dz * ((xa[u][j] - mu[j]) * rs[j]);
                }
        }
    }
    for (; row < r1; row += rpi) {
        float xv[VEC], gv[VEC];
        if (VEC == 4) {
            const float4 t = da_ldq(x, row * cq + q);
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            if (MODE == 2) {
                const float4 g = da_ldq(dy, row * cq + q);
                gv[0] = g.x; gv[1] = g.y; gv[2] = g.z; gv[3] = g.w;
            }
        } else {
            xv[0] = da_ld1(x, row * C + q);
            if (MODE == 2) gv[0] = da_ld1(dy, row * C + q);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (MODE == 0) { a0[j] += xv[j]; a1[j] += xv[j] * xv[j]; }
            else if (MODE == 1) { a0[j] += xv[j]; }
            else {
                const float z = xv[j] * sc[j] + sf[j];
                const float dz = gv[j] * da_act_grad(z, slope);
                const float xh = (xv[j] - mu[j]) * rs[j];
                a0[j] += dz; a1[j] += dz * xh;
            }
        }
    }
    double* s0 = sh;
    double* s1 = sh + (size_t)rpi * C;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s0[r * C + q * VEC + j] = (double)a0[j]; s1[r * C + q * VEC + j] = (double)a1[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double t0 = 0.0, t1 = 0.0;
        for (int rr = 0; rr < rpi; ++rr) { t0 += s0[rr * C + c]; t1 += s1[rr * C + c]; }
        partial[((size_t)blockIdx.x * 2 + 0) * C + c] = t0;
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = t1;
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ partial, int nblocks, long long M, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, float* running_mean, float* running_var,
                                   float* mean, float* rstd, float* scale, float* shift) {
    // one wave per channel: lanes stride over the block partials, then a shuffle reduction
    const int c = blockIdx.x;
    double s = 0.0, ss = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) { s += partial[((size_t)b * 2) * C + c]; ss += partial[((size_t)b * 2 + 1) * C + c]; }
    s = da_wave_sum(s); ss = da_wave_sum(ss);
    if (threadIdx.x != 0) return;
    const double m = s / (double)M;
    double var = ss / (double)M - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean[c] = mf; rstd[c] = rs;
    const float sc = g * rs;
    scale[c] = sc; shift[c] = b - mf * sc;
    if (running_mean) {
        const double unbiased = (M > 1) ? var * ((double)M / (double)(M - 1)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* mean, float* rstd, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rs = 1.f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean[c] = rm[c]; rstd[c] = rs; scale[c] = g * rs; shift[c] = b - rm[c] * g * rs;
}

template <int VEC, typename T = float>
__global__ void bn_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                  const float* __restrict__ shift, float slope, T* __restrict__ y,
                                  long long nvec, int cq) {
    // cq divides the block size for every channel count on the path (C/4 in {1,2,4,8,16}), so a thread always sees
    // the same channel quad: its constants are loaded once into registers
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool fixed_q = (VEC == 4) && (stride % cq == 0);
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sf = sc;
    if (fixed_q) { const int q = (int)(i0 % cq); sc = reinterpret_cast<const float4*>(scale)[q]; sf = reinterpret_cast<const float4*>(shift)[q]; }
    if (VEC == 4 && fixed_q && sizeof(T) == 4) {
        // four 16-byte loads in flight per lane before the first store; streaming (non-temporal) accesses: nothing here is read twice
        typedef float f4v __attribute__((ext_vector_type(4)));
        const float4* xi = reinterpret_cast<const float4*>(x); float4* yo = reinterpret_cast<float4*>(y);
        const f4v* xv = reinterpret_cast<const f4v*>(x); f4v* yv = reinterpret_cast<f4v*>(y);
        long long i = i0;
        for (; i + 3 * stride < nvec; i += 4 * stride) {
            f4v t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = __builtin_nontemporal_load(xv + i + k * stride);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f4v o;
                o.x = da_act(t[k].x * sc.x + sf.x, slope); o.y = da_act(t[k].y * sc.y + sf.y, slope);
                o.z = da_act(t[k].z * sc.z + sf.z, slope); o.w = da_act(t[k].w * sc.w + sf.w, slope);
                __builtin_nontemporal_store(o, yv + i + k * stride);
            }
        }
        for (; i < nvec; i += stride) {
            const float4 t = xi[i];
            float4 o;
            o.x = da_act(t.x * sc.x + sf.x, slope); o.y = da_act(t.y * sc.y + sf.y, slope);
            o.z = da_act(t.z * sc.z + sf.z, slope); o.w = da_act(t.w * sc.w + sf.w, slope);
            yo[i] = o;
        }
        return;
    }
    for (long long i = i0; i < nvec; i += stride) {
        if (VEC == 4) {
            if (!fixed_q) { const int q = (int)(i % cq); sc = reinterpret_cast<const float4*>(scale)[q]; sf = reinterpret_cast<const float4*>(shift)[q]; }
            const float4 t = da_ldq(x, i);
            float4 o;
            o.x = da_act(t.x * sc.x + sf.x, slope); o.y = da_act(t.y * sc.y + sf.y, slope);
            o.z = da_act(t.z * sc.z + sf.z, slope); o.w = da_act(t.w * sc.w + sf.w, slope);
            da_stq(y, i, o);
        } else {
            const int q = (int)(i % cq);
            da_st1(y, i, da_act(da_ld1(x, i) * scale[q] + shift[q], slope));
        }
    }
}

// finalize BN-backward sums: dgamma = s2, dbeta = s1, cm[0][c] = s1/M, cm[1][c] = s2/M
// `rstd_scale`: the partials hold sum dz (x - mean) instead of sum dz xhat (sums accumulated by the PRODUCER of dy, which has no rstd at hand)
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblocks, long long M, int C,
                                       float* dgamma, float* dbeta, float* cm, const float* __restrict__ rstd_scale = nullptr) {
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) { s1 += partial[((size_t)b * 2) * C + c]; s2 += partial[((size_t)b * 2 + 1) * C + c]; }
    s1 = da_wave_sum(s1); s2 = da_wave_sum(s2);
    if (threadIdx.x != 0) return;
    if (rstd_scale) s2 *= (double)rstd_scale[c];
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    cm[c] = (float)(s1 / (double)M);
    cm[C + c] = (float)(s2 / (double)M);
}

// <= 56 registers: two waves of this HBM-bound pass then fit per SIMD beside the two resident 200-register waves of the split weight gradient that
// the training step runs next to it on the side stream (conv3d_wgring.h); at 60 only one did
template <int VEC, typename T = float>
__global__ void bn_act_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                        const float* __restrict__ cm, float slope, int train,
                                        T* __restrict__ dx, long long nvec, int cq, int C, double* __restrict__ dxsum_partial) {
    // Three 16-byte streams (dy, x -> dx).  The six per-channel constants of this thread's channel quad are hoisted
    // into registers (the grid stride is a multiple of cq, so the quad never changes); UNR load pairs in flight.
    constexpr int UNR = (VEC == 4) ? 2 : 1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long i00 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool fixed_q = (stride % cq == 0);
    float k_sc[VEC], k_sf[VEC], k_mu[VEC], k_c1[VEC], k_c2[VEC];      // (k_c1 = scale mean(dz), k_c2 = scale rstd mean(dz xhat): five quads, not six -- the register budget above)
    auto load_consts = [&](int q) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = q * VEC + j;
            k_sc[j] = scale[c]; k_sf[j] = shift[c]; k_mu[j] = mean[c]; k_c1[j] = scale[c] * cm[c]; k_c2[j] = scale[c] * rstd[c] * cm[C + c];
        }
    };
    if (fixed_q) load_consts((int)(i00 % cq));
    float colacc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) colacc[j] = 0.f;
    for (long long i0 = i00; i0 < nvec; i0 += stride * UNR) {
        float xv[UNR][VEC], gv[UNR][VEC];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            // unconditional loads (a slot past the end re-reads the last one and stores nothing): a load under a branch is waited for where the branch
            // ends, and the pairs would not be in flight together
            const long long i = (i0 + u * stride < nvec) ? i0 + u * stride : nvec - 1;
            if (VEC == 4) {
                const float4 t = da_ldq_nt(x, i);
                const float4 g = da_ldq_nt(dy, i);
                xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][3] = t.w;
                gv[u][0] = g.x; gv[u][1] = g.y; gv[u][2] = g.z; gv[u][3] = g.w;
            } else { xv[u][0] = da_ld1(x, i); gv[u][0] = da_ld1(dy, i); }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long i = i0 + u * stride;
            if (i >= nvec) continue;
            if (!fixed_q) load_consts((int)(i % cq));
            float o[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float z = xv[u][j] * k_sc[j] + k_sf[j];
                const float dz = gv[u][j] * da_act_grad(z, slope);
                if (train) {
                    o[j] = k_sc[j] * dz - k_c1[j] - (xv[u][j] - k_mu[j]) * k_c2[j];      // = scale (dz - mean(dz) - xhat mean(dz xhat))
                } else {
                    o[j] = k_sc[j] * dz;
                }
            }
            if (VEC == 4) da_stq_nt(dx, i, make_float4(o[0], o[1], o[2], o[3]));
            else da_st1(dx, i, o[0]);
#pragma unroll
            for (int j = 0; j < VEC; ++j) colacc[j] += o[j];
        }
    }
    // optional: column sums of dx (= the bias gradient of the convolution that produced x), fused here so the conv's
    // weight-gradient call needs no separate pass over dy.  Only valid when the channel quad is fixed per thread.
    if (dxsum_partial != nullptr && fixed_q && VEC == 4) {
        __shared__ float shs[256 * 4];
#pragma unroll
        for (int j = 0; j < VEC; ++j) shs[threadIdx.x * 4 + j] = colacc[j];
        __syncthreads();
        const int q0 = (int)(((long long)blockIdx.x * blockDim.x) % cq);
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            // threads t with (q0 + t) % cq == c / 4 own channel c's quad
            const int want = c >> 2, j = c & 3;
            double t = 0.0;
            for (int th = ((want - q0) % cq + cq) % cq; th < (int)blockDim.x; th += cq) t += (double)shs[th * 4 + j];
            dxsum_partial[((size_t)blockIdx.x * 2) * C + c] = t;
        }
    }
}

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, float slope,
                               T* __restrict__ dx, long long n) {
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 g = da_ldq(dy, i);
        const float4 v = da_ldq(y, i);
        float4 o;
        o.x = g.x * (v.x > 0.f ? 1.f : slope); o.y = g.y * (v.y > 0.f ? 1.f : slope);
        o.z = g.z * (v.z > 0.f ? 1.f : slope); o.w = g.w * (v.w > 0.f ? 1.f : slope);
        da_stq(dx, i, o);
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        da_st1(dx, i, da_ld1(dy, i) * (da_ld1(y, i) > 0.f ? 1.f : slope));
}

// Backward of a conv block WITHOUT BatchNorm (modules.convBlock, the registration net): dx = (g1 [+ g2]) * act'(y) and, in the same
// pass, the per-channel column sums of dx = the convolution's bias gradient.  g2 is the second incoming gradient when the block's
// output has two consumers (skip connection): the sum autograd would form in its own pass happens here.  Row-blocked like
// col_partial_kernel: a thread keeps one channel quad, per-thread fp32 sums -> per-block doubles -> colsum_finalize_kernel.
template <int VEC, typename T = float>
__global__ void act_bwd_add_dbias_kernel(const T* __restrict__ g1, const T* __restrict__ g2, const T* __restrict__ y,
                                         float slope, T* __restrict__ dx, long long M, int C, long long rows_per_block,
                                         double* __restrict__ partial) {
    extern __shared__ double sh[];   // [rpi][C]
    const int cq = C / VEC;
    const int rpi = blockDim.x / cq;
    const int q = threadIdx.x % cq, r = threadIdx.x / cq;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    float a0[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) a0[j] = 0.f;
    for (long long row = r0 + r; row < r1; row += rpi) {
        float gv[VEC], yv[VEC];
        if (VEC == 4) {
            float4 t = da_ldq_nt(g1, row * cq + q);
            if (g2) { const float4 u = da_ldq_nt(g2, row * cq + q); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
            if (y) { const float4 v = da_ldq_nt(y, row * cq + q); yv[0] = v.x; yv[1] = v.y; yv[2] = v.z; yv[3] = v.w; }
        } else {
            gv[0] = da_ld1(g1, row * C + q) + (g2 ? da_ld1(g2, row * C + q) : 0.f);
            if (y) yv[0] = da_ld1(y, row * C + q);
        }
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { o[j] = y ? gv[j] * (yv[j] > 0.f ? 1.f : slope) : gv[j]; a0[j] += o[j]; }
        if (VEC == 4) da_stq_nt(dx, row * cq + q, make_float4(o[0], o[1], o[2], o[3]));
        else da_st1(dx, row * C + q, o[0]);
    }
    if (!partial) return;
#pragma unroll
    for (int j = 0; j < VEC; ++j) sh[r * C + q * VEC + j] = (double)a0[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double t0 = 0.0;
        for (int rr = 0; rr < rpi; ++rr) t0 += sh[rr * C + c];
        partial[((size_t)blockIdx.x * 2 + 0) * C + c] = t0;
    }
}

__global__ void colsum_finalize_kernel(const double* __restrict__ partial, int nblocks, int C, float* out, int accumulate) {
    const int c = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[((size_t)b * 2) * C + c];
    s = da_wave_sum(s);
    if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)s : (float)s;
}

template <int MODE, typename T = float>
int launch_partial(const RowPlan& p, const T* x, const T* dy, const float* mean, const float* rstd,
                   const float* scale, const float* shift, float slope, long long M, int C, double* partial, hipStream_t st) {
    const size_t shm = (size_t)2 * p.rpi * C * sizeof(double);
    if (p.vec == 4)
        hipLaunchKernelGGL((col_partial_kernel<4, MODE, T>), dim3(p.grid), dim3(p.block), shm, st, x, dy, mean, rstd, scale, shift, slope, M, C, p.rows_per_block, partial);
    else
        hipLaunchKernelGGL((col_partial_kernel<1, MODE, T>), dim3(p.grid), dim3(p.block), shm, st, x, dy, mean, rstd, scale, shift, slope, M, C, p.rows_per_block, partial);
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" size_t da_bn_ws_bytes(long long M, int C) {
    (void)M;
    return da_align((size_t)kMaxBlocks * 2 * C * sizeof(double)) + da_align((size_t)2 * C * sizeof(float));
}

template <typename T>
static int bn_train_stats_t(const T* x, long long M, int C, const float* gamma, const float* beta,
                            float eps, float momentum, float* running_mean, float* running_var,
                            float* mean, float* rstd, float* scale, float* shift,
                            void* ws, size_t ws_bytes, void* stream) {
    if (!x || M <= 0 || C <= 0 || C > 1024) return DA_ERR_BADARG;
    if (ws_bytes < da_bn_ws_bytes(M, C)) return DA_ERR_WS_SMALL;
    const RowPlan p = plan_rows(M, C);
    double* partial = (double*)ws;
    int rc = launch_partial<0, T>(p, x, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, M, C, partial, da_stream(stream));
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, da_stream(stream), partial, p.grid, M, C,
                       gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift);
    DA_LAUNCH_CHECK();
    return 0;
}
extern "C" int da_bn_train_stats(const float* x, long long M, int C, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var,
                                 float* mean, float* rstd, float* scale, float* shift,
                                 void* ws, size_t ws_bytes, void* stream) {
    return bn_train_stats_t<float>(x, M, C, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift, ws, ws_bytes, stream);
}
extern "C" int da_bn_train_stats_bf16(const void* x, long long M, int C, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var,
                                      float* mean, float* rstd, float* scale, float* shift,
                                      void* ws, size_t ws_bytes, void* stream) {
    return bn_train_stats_t<da_bf16>((const da_bf16*)x, M, C, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift, ws, ws_bytes, stream);
}

extern "C" int da_bn_train_stats_from_partials(const double* partial, int nparts, long long M, int C,
                                               const float* gamma, const float* beta, float eps, float momentum,
                                               float* running_mean, float* running_var,
                                               float* mean, float* rstd, float* scale, float* shift, void* stream) {
    if (!partial || nparts <= 0 || M <= 0 || C <= 0 || C > 1024) return DA_ERR_BADARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, da_stream(stream), partial, nparts, M, C,
                       gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                 float eps, int C, float* mean, float* rstd, float* scale, float* shift, void* stream) {
    if (!running_mean || !running_var || C <= 0) return DA_ERR_BADARG;
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(da_cdiv(C, 64)), dim3(64), 0, da_stream(stream), gamma, beta, running_mean, running_var, eps, C, mean, rstd, scale, shift);
    DA_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int bn_act_fwd_t(const T* x, const float* scale, const float* shift, float act_slope, T* y, long long M, int C, void* stream) {
    if (!x || !y || M <= 0 || C <= 0) return DA_ERR_BADARG;
    if (C % 4 == 0) {
        const long long nvec = M * C / 4;
        hipLaunchKernelGGL((bn_act_fwd_kernel<4, T>), dim3(da_grid(nvec, 256)), dim3(256), 0, da_stream(stream), x, scale, shift, act_slope, y, nvec, C / 4);
    } else {
        const long long nvec = M * C;
        hipLaunchKernelGGL((bn_act_fwd_kernel<1, T>), dim3(da_grid(nvec, 256)), dim3(256), 0, da_stream(stream), x, scale, shift, act_slope, y, nvec, C);
    }
    DA_LAUNCH_CHECK();
    return 0;
}
extern "C" int da_bn_act_fwd(const float* x, const float* scale, const float* shift, float act_slope, float* y,
                             long long M, int C, void* stream) {
    return bn_act_fwd_t<float>(x, scale, shift, act_slope, y, M, C, stream);
}
extern "C" int da_bn_act_fwd_bf16(const void* x, const float* scale, const float* shift, float act_slope, void* y,
                                  long long M, int C, void* stream) {
    return bn_act_fwd_t<da_bf16>((const da_bf16*)x, scale, shift, act_slope, (da_bf16*)y, M, C, stream);
}

template <typename T>
static int colsum_t(const T* x, long long M, int C, float* out, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !out || M <= 0 || C <= 0 || C > 1024) return DA_ERR_BADARG;
    if (ws_bytes < da_bn_ws_bytes(M, C)) return DA_ERR_WS_SMALL;
    const RowPlan p = plan_rows(M, C);
    double* partial = (double*)ws;
    int rc = launch_partial<1, T>(p, x, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, M, C, partial, da_stream(stream));
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C), dim3(64), 0, da_stream(stream), partial, p.grid, C, out, 0);
    DA_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int bn_act_bwd_impl(const T* dy, const T* x, const float* mean, const float* rstd,
                           const float* scale, const float* shift, float act_slope, int train,
                           T* dx, float* dgamma, float* dbeta, float* dxsum, long long M, int C,
                           void* ws, size_t ws_bytes, void* stream, const double* pre = nullptr, int pre_n = 0) {
    if (!dy || !x || !dx || M <= 0 || C <= 0 || C > 1024) return DA_ERR_BADARG;
    if (ws_bytes < da_bn_ws_bytes(M, C)) return DA_ERR_WS_SMALL;
    const RowPlan p = plan_rows(M, C);
    double* partial = (double*)ws;
    float* cm = (float*)((char*)ws + da_align((size_t)kMaxBlocks * 2 * C * sizeof(double)));
    hipStream_t st = da_stream(stream);
    if (pre && pre_n > 0) {
        // the two sums arrive from the kernel that produced dy (sum dz, sum dz (x - mean) per workgroup): no reduction pass over (dy, x)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, pre, pre_n, M, C, dgamma, dbeta, cm, rstd);
    } else {
        int rc = launch_partial<2, T>(p, x, dy, mean, rstd, scale, shift, act_slope, M, C, partial, st);
        if (rc) return rc;
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, partial, p.grid, M, C, dgamma, dbeta, cm, (const float*)nullptr);
    }
    DA_LAUNCH_CHECK();
    if (C % 4 == 0) {
        const long long nvec = M * C / 4;
        // fused bias-gradient column sums need a fixed channel quad per thread (256 % (C/4) == 0) and <= kMaxBlocks blocks
        const int cq = C / 4;
        const bool fuse = dxsum != nullptr && (256 % cq) == 0;
        const int grid = fuse ? da_grid(nvec, 256, kMaxBlocks) : da_grid(nvec, 256);
        hipLaunchKernelGGL((bn_act_bwd_apply_kernel<4, T>), dim3(grid), dim3(256), 0, st, dy, x, mean, rstd, scale, shift, cm, act_slope, train, dx, nvec, cq, C, fuse ? partial : nullptr);
        DA_LAUNCH_CHECK();
        if (fuse) {
            hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C), dim3(64), 0, st, partial, grid, C, dxsum, 0);
            DA_LAUNCH_CHECK();
        } else if (dxsum != nullptr) {
            return colsum_t<T>(dx, M, C, dxsum, ws, ws_bytes, stream);
        }
    } else {
        const long long nvec = M * C;
        hipLaunchKernelGGL((bn_act_bwd_apply_kernel<1, T>), dim3(da_grid(nvec, 256)), dim3(256), 0, st, dy, x, mean, rstd, scale, shift, cm, act_slope, train, dx, nvec, C, C, nullptr);
        DA_LAUNCH_CHECK();
        if (dxsum != nullptr) return colsum_t<T>(dx, M, C, dxsum, ws, ws_bytes, stream);
    }
    return 0;
}

// Internal (pointwise_internal.h): only the SUMS of a training-mode BatchNorm + activation backward -- dgamma, dbeta and cm[2][C] = (mean dz, mean dz xhat),
// the constants bn_act_bwd_apply_kernel takes -- for a consumer that applies them itself (the fused transposed-conv backward, pointwise_mfma.hip).
// pre / pre_n: sums a producer of dy accumulated (da_bn_act_bwd_dbias_pre); otherwise one reduction pass over (dy, x).  cm lies inside ws (returned).
int da_bn_bwd_sums(const float* dy, const float* x, const float* mean, const float* rstd, const float* scale, const float* shift, float act_slope,
                   long long M, int C, float* dgamma, float* dbeta, const float** cm_out, void* ws, size_t ws_bytes, hipStream_t st, const double* pre, int pre_n) {
    if (!dy || !x || M <= 0 || C <= 0 || C > 1024) return DA_ERR_BADARG;
    if (ws_bytes < da_bn_ws_bytes(M, C)) return DA_ERR_WS_SMALL;
    const RowPlan p = plan_rows(M, C);
    double* partial = (double*)ws;
    float* cm = (float*)((char*)ws + da_align((size_t)kMaxBlocks * 2 * C * sizeof(double)));
    if (pre && pre_n > 0) {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, pre, pre_n, M, C, dgamma, dbeta, cm, rstd);
    } else {
        int rc = launch_partial<2, float>(p, x, dy, mean, rstd, scale, shift, act_slope, M, C, partial, st);
        if (rc) return rc;
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, partial, p.grid, M, C, dgamma, dbeta, cm, (const float*)nullptr);
    }
    DA_LAUNCH_CHECK();
    *cm_out = cm;
    return 0;
}

extern "C" int da_bn_act_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                             const float* scale, const float* shift, float act_slope, int train,
                             float* dx, float* dgamma, float* dbeta, long long M, int C,
                             void* ws, size_t ws_bytes, void* stream) {
    (void)gamma;
    return bn_act_bwd_impl<float>(dy, x, mean, rstd, scale, shift, act_slope, train, dx, dgamma, dbeta, nullptr, M, C, ws, ws_bytes, stream);
}
extern "C" int da_bn_act_bwd_dbias(const float* dy, const float* x, const float* mean, const float* rstd,
                                   const float* scale, const float* shift, float act_slope, int train,
                                   float* dx, float* dgamma, float* dbeta, float* dxsum, long long M, int C,
                                   void* ws, size_t ws_bytes, void* stream) {
    return bn_act_bwd_impl<float>(dy, x, mean, rstd, scale, shift, act_slope, train, dx, dgamma, dbeta, dxsum, M, C, ws, ws_bytes, stream);
}
// da_bn_act_bwd_dbias with the reduction pass replaced by sums the producer of dy accumulated in its epilogue: pre[pre_n][2][C] doubles =
// (sum dz, sum dz (x - mean)) with dz = dy act'(x scale + shift) (da_head_dice_bwd_bst; autograd of unets.py:31-32)
extern "C" int da_bn_act_bwd_dbias_pre(const float* dy, const float* x, const float* mean, const float* rstd,
                                       const float* scale, const float* shift, float act_slope, int train,
                                       float* dx, float* dgamma, float* dbeta, float* dxsum, long long M, int C,
                                       const double* pre, int pre_n, void* ws, size_t ws_bytes, void* stream) {
    if (!pre || pre_n <= 0 || !train) return DA_ERR_BADARG;
    return bn_act_bwd_impl<float>(dy, x, mean, rstd, scale, shift, act_slope, train, dx, dgamma, dbeta, dxsum, M, C, ws, ws_bytes, stream, pre, pre_n);
}
extern "C" int da_bn_act_bwd_dbias_bf16(const void* dy, const void* x, const float* mean, const float* rstd,
                                        const float* scale, const float* shift, float act_slope, int train,
                                        void* dx, float* dgamma, float* dbeta, float* dxsum, long long M, int C,
                                        void* ws, size_t ws_bytes, void* stream) {
    return bn_act_bwd_impl<da_bf16>((const da_bf16*)dy, (const da_bf16*)x, mean, rstd, scale, shift, act_slope, train, (da_bf16*)dx, dgamma, dbeta, dxsum, M, C, ws, ws_bytes, stream);
}

template <typename T>
static int act_bwd_t(const T* dy, const T* y, float act_slope, T* dx, long long numel, void* stream) {
    if (!dy || !y || !dx || numel <= 0) return DA_ERR_BADARG;
    hipLaunchKernelGGL((act_bwd_kernel<T>), dim3(da_grid(numel / 4 + 1, 256)), dim3(256), 0, da_stream(stream), dy, y, act_slope < 0.f ? 1.f : act_slope, dx, numel);
    DA_LAUNCH_CHECK();
    return 0;
}
extern "C" int da_act_bwd(const float* dy, const float* y, float act_slope, float* dx, long long numel, void* stream) {
    return act_bwd_t<float>(dy, y, act_slope, dx, numel, stream);
}
extern "C" int da_act_bwd_bf16(const void* dy, const void* y, float act_slope, void* dx, long long numel, void* stream) {
    return act_bwd_t<da_bf16>((const da_bf16*)dy, (const da_bf16*)y, act_slope, (da_bf16*)dx, numel, stream);
}

// `partial` == nullptr: no bias gradient.  Returns the number of per-block partials through *nparts (when given).
template <typename T>
static int act_bwd_add_t(const T* g1, const T* g2, const T* y, float act_slope, T* dx, long long M, int C, double* partial, int* nparts, void* stream) {
    if (!g1 || !dx || M <= 0 || C <= 0 || C > 1024) return DA_ERR_BADARG;
    const RowPlan p = plan_rows(M, C);
    const T* yy = act_slope < 0.f ? nullptr : y;          // no activation: dx = g1 + g2
    if (act_slope >= 0.f && !y) return DA_ERR_BADARG;
    const size_t shm = (size_t)p.rpi * C * sizeof(double);
    hipStream_t st = da_stream(stream);
    if (p.vec == 4) hipLaunchKernelGGL((act_bwd_add_dbias_kernel<4, T>), dim3(p.grid), dim3(p.block), shm, st, g1, g2, yy, act_slope, dx, M, C, p.rows_per_block, partial);
    else hipLaunchKernelGGL((act_bwd_add_dbias_kernel<1, T>), dim3(p.grid), dim3(p.block), shm, st, g1, g2, yy, act_slope, dx, M, C, p.rows_per_block, partial);
    DA_LAUNCH_CHECK();
    if (nparts) *nparts = p.grid;
    return 0;
}
template <typename T>
static int act_bwd_add_dbias_t(const T* g1, const T* g2, const T* y, float act_slope, T* dx, float* dbias,
                               long long M, int C, void* ws, size_t ws_bytes, void* stream) {
    if (M <= 0 || C <= 0 || C > 1024) return DA_ERR_BADARG;
    if (dbias && ws_bytes < da_bn_ws_bytes(M, C)) return DA_ERR_WS_SMALL;
    int np = 0;
    const int rc = act_bwd_add_t<T>(g1, g2, y, act_slope, dx, M, C, dbias ? (double*)ws : nullptr, &np, stream);
    if (rc) return rc;
    if (dbias) {
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C), dim3(64), 0, da_stream(stream), (const double*)ws, np, C, dbias, 0);
        DA_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int da_act_bwd_add_dbias(const float* g1, const float* g2, const float* y, float act_slope, float* dx, float* dbias,
                                    long long M, int C, void* ws, size_t ws_bytes, void* stream) {
    return act_bwd_add_dbias_t<float>(g1, g2, y, act_slope, dx, dbias, M, C, ws, ws_bytes, stream);
}
extern "C" int da_act_bwd_add_dbias_bf16(const void* g1, const void* g2, const void* y, float act_slope, void* dx, float* dbias,
                                         long long M, int C, void* ws, size_t ws_bytes, void* stream) {
    return act_bwd_add_dbias_t<da_bf16>((const da_bf16*)g1, (const da_bf16*)g2, (const da_bf16*)y, act_slope, (da_bf16*)dx, dbias, M, C, ws, ws_bytes, stream);
}

// The two halves of da_act_bwd_add_dbias as separate entries, for callers that keep the tiny per-channel finish off the stream the big
// pass runs on (ops.py queues it on the weight-gradient side stream: a 32-workgroup kernel that must wait for a free CU slot behind
// persistent matrix kernels costs the main stream 20 - 100 us per layer): `partial` is caller-owned, da_bn_ws_bytes(M, C) bytes.
extern "C" int da_act_bwd_add_partial(const float* g1, const float* g2, const float* y, float act_slope, float* dx,
                                      long long M, int C, void* partial, size_t partial_bytes, int* nparts, void* stream) {
    if (!partial || !nparts || M <= 0 || C <= 0 || C > 1024) return DA_ERR_BADARG;
    if (partial_bytes < da_bn_ws_bytes(M, C)) return DA_ERR_WS_SMALL;
    return act_bwd_add_t<float>(g1, g2, y, act_slope, dx, M, C, (double*)partial, nparts, stream);
}
extern "C" int da_act_bwd_add_partial_bf16(const void* g1, const void* g2, const void* y, float act_slope, void* dx,
                                           long long M, int C, void* partial, size_t partial_bytes, int* nparts, void* stream) {
    if (!partial || !nparts || M <= 0 || C <= 0 || C > 1024) return DA_ERR_BADARG;
    if (partial_bytes < da_bn_ws_bytes(M, C)) return DA_ERR_WS_SMALL;
    return act_bwd_add_t<da_bf16>((const da_bf16*)g1, (const da_bf16*)g2, (const da_bf16*)y, act_slope, (da_bf16*)dx, M, C, (double*)partial, nparts, stream);
}

extern "C" int da_colsum_finish(const void* partial, int nparts, int C, float* out, int accumulate, void* stream) {
    if (!partial || !out || nparts <= 0 || C <= 0 || C > 1024) return DA_ERR_BADARG;
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(C), dim3(64), 0, da_stream(stream), (const double*)partial, nparts, C, out, accumulate);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_colsum(const float* x, long long M, int C, float* out, void* ws, size_t ws_bytes, void* stream) {
    return colsum_t<float>(x, M, C, out, ws, ws_bytes, stream);
}
extern "C" int da_colsum_bf16(const void* x, long long M, int C, float* out, void* ws, size_t ws_bytes, void* stream) {
    return colsum_t<da_bf16>((const da_bf16*)x, M, C, out, ws, ws_bytes, stream);
}
