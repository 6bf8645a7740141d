// Micro-benchmark behind the "split" matrix mode: fp32 products from three bf16 terms per operand.
//   a = a1 + a2 + a3 exactly (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 3 x 8 = 24 significand bits),
//   a*b ~= a1b1 + (a1b2 + a2b1) + (a2b2 + a1b3 + a3b1)   -- the three dropped terms are <= 2^-25 |ab| together --
//   six v_mfma_f32_16x16x32_bf16 (fp32 accumulate) instead of eight v_mfma_f32_16x16x4_f32 for the same 16x16x32 MACs.
// part 1 (accuracy): C = A B, M = N = 16, K = 448 (the conv's 27 taps x 16 cin padded to 14 K-steps of 32), 1024 random
//   problems; error against a double-precision host reference of (a) the native fp32 MFMA chain, (b) the split, (c) plain bf16.
// part 2 (rate): sustained MFMA issue of the split's instruction mix (6 MFMAs per M-tile and K-step, optionally with the three
//   ds_read_b128 that feed them).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/split_bf16.hip -o /tmp/split_bf16 ; run: /tmp/split_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf_round(float v) { return (float)(__bf16)v; }

constexpr int KS = 14, K = 32 * KS;

// A[prob][16][K], B[prob][K][16] -> C[prob][3][16][16]
__global__ void __launch_bounds__(64) acc_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int scale_mode) {
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const float* a = A + (size_t)blockIdx.x * 16 * K;
    const float* b = B + (size_t)blockIdx.x * K * 16;
    f32x4 c32 = {0.f, 0.f, 0.f, 0.f}, csp = c32, cbf = c32;
    for (int s = 0; s < KS; ++s) {
        // native fp32: eight 16x16x4 steps; lane supplies A[i][k], B[k][i] with k = 32 s + 4 m + g
        for (int m = 0; m < 8; ++m) {
            const int k = 32 * s + 4 * m + g;
            c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i * K + k], b[k * 16 + i], c32, 0, 0, 0);
        }
        bf16x8 a1, a2, a3, b1, b2, b3;
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * s + 8 * g + e;
            const float av = a[i * K + k], bv = b[k * 16 + i];
            const float ah = bf_round(av), am = bf_round(av - ah), al = bf_round(av - ah - am);
            const float bh = bf_round(bv), bm = bf_round(bv - bh), bl = bf_round(bv - bh - bm);
            a1[e] = (__bf16)ah; a2[e] = (__bf16)am; a3[e] = (__bf16)al;
            b1[e] = (__bf16)bh; b2[e] = (__bf16)bm; b3[e] = (__bf16)bl;
        }
        // small terms first
        csp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, csp, 0, 0, 0);
        csp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, csp, 0, 0, 0);
        csp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, csp, 0, 0, 0);
        csp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, csp, 0, 0, 0);
        csp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, csp, 0, 0, 0);
        csp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, csp, 0, 0, 0);
        cbf = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, cbf, 0, 0, 0);
    }
    float* c = C + (size_t)blockIdx.x * 3 * 256;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        c[0 * 256 + row * 16 + i] = c32[r];
        c[1 * 256 + row * 16 + i] = csp[r];
        c[2 * 256 + row * 16 + i] = cbf[r];
    }
}

// rate: VAR 0 = fp32 16x16x4 (8 M-tiles, 4 per step); 1 = split, MFMAs only; 2 = split + 3 ds_read_b128 per (M-tile, step); 3 = plain bf16 + 1 read
template <int VAR>
__global__ void __launch_bounds__(256, 2) rate_kernel(float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    for (int t = threadIdx.x; t < 3 * 8192; t += 256) lds[t] = (float)(t & 7) * 0.001f;
    __syncthreads();
    f32x4 acc[8];
    for (int r = 0; r < 8; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* abase = lds + ((wave * 180 + i) * 8 + (g & 1) * 4);       // bf16 tile: 16 cin = 32 B = 8 floats per voxel
    if constexpr (VAR == 0) {
        f32x4 a[8]; for (int r = 0; r < 8; ++r) a[r] = (f32x4){1.f + r, 0.5f, 0.25f, 0.125f};
        f32x4 b = {0.1f, 0.2f, 0.3f, 0.4f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][m], b[m], acc[r], 0, 0, 0);
        }
    } else {
        bf16x8 b1, b2, b3;
        for (int e = 0; e < 8; ++e) { b1[e] = (__bf16)(0.1f * e); b2[e] = (__bf16)(0.001f * e); b3[e] = (__bf16)(0.00001f * e); }
        bf16x8 a1[8], a2[8], a3[8];
        for (int r = 0; r < 8; ++r) for (int e = 0; e < 8; ++e) { a1[r][e] = (__bf16)(1.f + r); a2[r][e] = (__bf16)0.01f; a3[r][e] = (__bf16)0.0001f; }
        for (int it = 0; it < iters; ++it) {
            if constexpr (VAR == 2 || VAR == 3) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float* p = abase + (r * 18 + (it % 3)) * 8;
                    a1[r] = *reinterpret_cast<const bf16x8*>(p);
                    if constexpr (VAR == 2) { a2[r] = *reinterpret_cast<const bf16x8*>(p + 8192); a3[r] = *reinterpret_cast<const bf16x8*>(p + 16384); }
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if constexpr (VAR == 3) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[r], b1, acc[r], 0, 0, 0);
            }
            if constexpr (VAR != 3) {
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[r], b3, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[r], b1, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[r], b2, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[r], b2, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[r], b1, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[r], b1, acc[r], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 8; ++r) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


// The conv kernel's K loop in split mode (CK = 8: a voxel's 8 cin = one 16-byte bf16x8 fragment; lane group g = one of the 4 taps of a K-step).
// LDS: three planes [6][10][18][8] bf16 (17 280 B each).  PIPE 4: per K-step 8 rows x 3 planes, fragments of the next 2 rows read while the
// 12 MFMAs of the current 2 rows issue.  PIPE 6: halo-row reuse -- the 4 taps of a class share dy, so the fragment of halo row h serves
// (r, dy) = (h, 0), (h - 1, 1), (h - 2, 2): 10 row fragments per class instead of 24 (2 classes + one leftover step per chunk).
struct F3 { bf16x8 p[3]; };
template <int PIPE>
__global__ void __launch_bounds__(256, 2) loop_kernel(float* __restrict__ out, int items) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    constexpr int PLANE = 6 * 10 * 18 * 16;     // bytes
    for (int t = threadIdx.x; t < 3 * PLANE / 4; t += 256) lds[t] = (float)(t & 7) * 0.001f;
    __syncthreads();
    f32x4 acc[8];
    for (int r = 0; r < 8; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    F3 w[3];
    for (int d = 0; d < 3; ++d) for (int q = 0; q < 3; ++q) for (int e = 0; e < 8; ++e) w[d].p[q][e] = (__bf16)(0.1f * (e + d) / (1 + 100 * q));
    const char* base = reinterpret_cast<const char*>(lds) + ((wave * 10) * 18 + i) * 16;
    auto ld = [&](int off) -> F3 {
        F3 f; const char* a = base + off;
        f.p[0] = *reinterpret_cast<const bf16x8*>(a); f.p[1] = *reinterpret_cast<const bf16x8*>(a + PLANE); f.p[2] = *reinterpret_cast<const bf16x8*>(a + 2 * PLANE);
        return f;
    };
    auto mm = [&](f32x4& c0, f32x4& c1, const F3& a0, const F3& a1, const F3& b) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[0], b.p[2], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.p[0], b.p[2], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[2], b.p[0], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.p[2], b.p[0], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[1], b.p[1], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.p[1], b.p[1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[0], b.p[1], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.p[0], b.p[1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[1], b.p[0], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.p[1], b.p[0], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[0], b.p[0], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.p[0], b.p[0], c1, 0, 0, 0);
    };
    auto mm1 = [&](f32x4& c0, const F3& a0, const F3& b) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[0], b.p[2], c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[2], b.p[0], c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[1], b.p[1], c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[0], b.p[1], c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[1], b.p[0], c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[0], b.p[0], c0, 0, 0, 0);
    };
    const int tapoff = ((g >> 1) * 10 * 18 + (g & 1)) * 16;       // lane group -> (dz, dx) of its tap
    for (int it = 0; it < items; ++it) {
        if constexpr (PIPE == 4) {
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                const int so = tapoff + (s % 3) * 288 + (s / 3) * 16;
                F3 a0 = ld(so), a1 = ld(so + 288);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    F3 n0, n1;
                    if (q < 3) { n0 = ld(so + (2 * q + 2) * 288); n1 = ld(so + (2 * q + 3) * 288); }
                    mm(acc[2 * q], acc[2 * q + 1], a0, a1, w[s % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q < 3) { a0 = n0; a1 = n1; }
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int so = tapoff + c * 32;
                F3 f = ld(so);
#pragma unroll
                for (int h = 0; h < 10; ++h) {
                    F3 fn; if (h < 9) fn = ld(so + (h + 1) * 288);
                    // (r, dy) with r + dy = h: three independent accumulators, interleaved product by product
                    if (h >= 2 && h <= 7) {
#pragma unroll
                        for (int pr = 0; pr < 6; ++pr) {
                            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy)
                                acc[h - dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.p[PA[pr]], w[dy].p[PB[pr]], acc[h - dy], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) { const int r = h - dy; if (r >= 0 && r < 8) mm1(acc[r], f, w[dy]); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (h < 9) f = fn;
                }
            }
            {   // leftover step: lane groups 0..2 = dy 0..2 of the ninth (dz, dx), group 3 idle (zero weights)
                const int so = ((g < 3 ? g : 0) * 18 + 2) * 16 + 2 * 10 * 18 * 16;
                F3 a0 = ld(so), a1 = ld(so + 288);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    F3 n0, n1;
                    if (q < 3) { n0 = ld(so + (2 * q + 2) * 288); n1 = ld(so + (2 * q + 3) * 288); }
                    mm(acc[2 * q], acc[2 * q + 1], a0, a1, w[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q < 3) { a0 = n0; a1 = n1; }
                }
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 8; ++r) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int PIPE>
static void run_loop(const char* name) {
    float* out; hipMalloc(&out, 512 * 256 * sizeof(float));
    const int items = 3000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 17280);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop_kernel<PIPE>, dim3(512), dim3(256), 3 * 17280, 0, out, items);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 56 * 16 * 16 * 32 * (double)items * 512 * 4;      // 56 (row, K-step) pairs of 16x16x32 MACs per item and wave
    printf("%-44s %8.3f ms  %8.1f fp32-equivalent TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}

template <int VAR>
static void run_rate(const char* name, double macs_per_iter_per_wave) {
    float* out; hipMalloc(&out, 512 * 256 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<VAR>, dim3(512), dim3(256), 3 * 32768, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * macs_per_iter_per_wave * iters * 512 * 4;
    printf("%-44s %8.3f ms  %8.1f fp32-equivalent TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    const int P = 1024;
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<float> A((size_t)P * 16 * K), B((size_t)P * K * 16);
        srand(1234 + mode);
        auto rnd = [&]() { float u = 0.f; for (int t = 0; t < 12; ++t) u += (float)rand() / RAND_MAX; return u - 6.f; };   // ~N(0,1)
        for (auto& v : A) { v = rnd(); if (mode == 1) v = fabsf(v) + 0.5f; if (mode == 2) v *= expf(4.f * rnd()); }
        for (auto& v : B) { v = rnd(); if (mode == 1) v = fabsf(v) + 0.5f; if (mode == 2) v *= expf(4.f * rnd()); }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)P * 3 * 256 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(acc_kernel, dim3(P), dim3(64), 0, 0, dA, dB, dC, mode);
        std::vector<float> C((size_t)P * 3 * 256);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double num[3] = {0, 0, 0}, den = 0, mx[3] = {0, 0, 0}, mag = 0;
        for (int p = 0; p < P; ++p)
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double ref = 0, absref = 0;
                    for (int k = 0; k < K; ++k) { const double t = (double)A[((size_t)p * 16 + i) * K + k] * (double)B[((size_t)p * K + k) * 16 + j]; ref += t; absref += fabs(t); }
                    den += ref * ref; mag += absref;
                    for (int v = 0; v < 3; ++v) {
                        const double d = (double)C[((size_t)p * 3 + v) * 256 + i * 16 + j] - ref;
                        num[v] += d * d;
                        const double rel = fabs(d) / absref; if (rel > mx[v]) mx[v] = rel;       // error relative to sum |a_k b_k| (the condition-free bound)
                    }
                }
        const char* mn[3] = {"N(0,1) x N(0,1)", "positive operands (no cancellation)", "log-normal magnitudes (exp(4 N))"};
        printf("accuracy, %s, K = %d, %d problems of 16 x 16:\n", mn[mode], K, P);
        const char* vn[3] = {"fp32 MFMA chain (16x16x4)", "split 3 x bf16, 6 products", "plain bf16"};
        for (int v = 0; v < 3; ++v) printf("   %-30s rel-l2 %.3e   max |err| / sum|a b| %.3e\n", vn[v], sqrt(num[v] / den), mx[v]);
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    run_rate<0>("fp32 16x16x4, 32 MFMA / iter", 8.0 * 16 * 16 * 16);
    run_rate<1>("split: 48 x 16x16x32 bf16 / iter", 8.0 * 16 * 16 * 32);
    run_rate<2>("split + 24 ds_read_b128 / iter", 8.0 * 16 * 16 * 32);
    run_rate<3>("plain bf16: 8 MFMA + 8 ds_read_b128 / iter", 8.0 * 16 * 16 * 32);
    run_loop<4>("conv K loop, split, 2-row pipeline (168 rd)");
    run_loop<6>("conv K loop, split, halo-row reuse (84 rd)");
    return 0;
}
