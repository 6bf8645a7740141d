// What the matrix pipe sustains under RANDOM operands (power / DVFS, MI355X_MICROARCH.md "DVFS give-back"): rate and sustained shader clock of
//   0  v_mfma_f32_16x16x4_f32                       (fp32 matrix mode)
//   1  v_mfma_f32_16x16x32_bf16, 8 accumulators     (split mode's instruction)
//   2  v_mfma_f32_32x32x16_bf16, 4 accumulators
//   3  variant 1 fed like split mode: three operand planes whose magnitudes fall by 2^-8 per plane (h, m, l), six products per K-step
//   4  variant 3 + the three ds_read_b128 per (row, K-step) of the conv kernel's LDS tile (random contents), 2-row pipeline
//   5  variant 4 with halo-row reuse (half the LDS reads)
// Every kernel runs ~10 ms; clock = shader cycles (s_memtime) / wall (s_memrealtime, 100 MHz) of one workgroup.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o tools/ubench/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned rng(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ bf16x8 rand_frag(unsigned& s, int shift_exp) {      // 8 random bf16 in [1, 2) * 2^-shift_exp, random signs
    bf16x8 f;
    for (int e = 0; e < 8; ++e) {
        const unsigned r = rng(s) >> 8;
        const unsigned short bits = (unsigned short)(((r >> 7) & 1u) << 15 | ((127u - shift_exp) << 7) | (r & 127u));
        f[e] = __builtin_bit_cast(__bf16, bits);
    }
    return f;
}

struct F3 { bf16x8 p[3]; };

template <int VAR>
__global__ void __launch_bounds__(256, 2) k(float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    unsigned seed = blockIdx.x * 256 + threadIdx.x + 12345u;
    constexpr int PLANE = 6 * 10 * 18 * 16;     // bytes
    if (VAR >= 4) {
        unsigned short* l16 = reinterpret_cast<unsigned short*>(lds);
        for (int t = threadIdx.x; t < 3 * PLANE / 2; t += 256) {
            const int plane = t / (PLANE / 2);
            const unsigned r = rng(seed) >> 8;
            l16[t] = (unsigned short)(((r >> 7) & 1u) << 15 | ((127u - 8 * plane) << 7) | (r & 127u));
        }
        __syncthreads();
    }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if constexpr (VAR == 0) {
        f32x4 acc[8]; for (int r = 0; r < 8; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 a[8], b[2];
        for (int r = 0; r < 8; ++r) for (int m = 0; m < 4; ++m) a[r][m] = __uint_as_float((rng(seed) & 0x807FFFFFu) | 0x3F800000u);
        for (int r = 0; r < 2; ++r) for (int m = 0; m < 4; ++m) b[r][m] = __uint_as_float((rng(seed) & 0x807FFFFFu) | 0x3F000000u);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][m], b[(r + m) & 1][m], acc[r], 0, 0, 0);
        }
        for (int r = 0; r < 8; ++r) sum += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    } else if constexpr (VAR == 1 || VAR == 3) {
        f32x4 acc[8]; for (int r = 0; r < 8; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        F3 a[8], b;
        for (int r = 0; r < 8; ++r) for (int pl = 0; pl < 3; ++pl) a[r].p[pl] = rand_frag(seed, VAR == 3 ? 8 * pl : 0);
        for (int pl = 0; pl < 3; ++pl) b.p[pl] = rand_frag(seed, VAR == 3 ? 8 * pl + 3 : 3);
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r].p[PA[pr]], b.p[PB[pr]], acc[r], 0, 0, 0);
        }
        for (int r = 0; r < 8; ++r) sum += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    } else if constexpr (VAR == 2) {
        f32x16 acc[4]; for (int r = 0; r < 4; ++r) for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
        bf16x8 a[4][3], b[3];
        for (int r = 0; r < 4; ++r) for (int pl = 0; pl < 3; ++pl) a[r][pl] = rand_frag(seed, 0);
        for (int pl = 0; pl < 3; ++pl) b[pl] = rand_frag(seed, 3);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r][pr % 3], b[(pr + r) % 3], acc[r], 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) for (int e = 0; e < 16; ++e) sum += acc[r][e];
    } else {
        f32x4 acc[8]; for (int r = 0; r < 8; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        F3 w[3];
        for (int d = 0; d < 3; ++d) for (int pl = 0; pl < 3; ++pl) w[d].p[pl] = rand_frag(seed, 8 * pl + 3);
        const char* base = reinterpret_cast<const char*>(lds) + ((wave * 10) * 18 + i) * 16;
        auto ld = [&](int off) -> F3 {
            F3 f; const char* a = base + off;
            f.p[0] = *reinterpret_cast<const bf16x8*>(a); f.p[1] = *reinterpret_cast<const bf16x8*>(a + PLANE); f.p[2] = *reinterpret_cast<const bf16x8*>(a + 2 * PLANE);
            return f;
        };
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
        const int tapoff = ((g >> 1) * 10 * 18 + (g & 1)) * 16;
        for (int it = 0; it < iters; ++it) {
            if constexpr (VAR == 4) {
#pragma unroll
                for (int s = 0; s < 7; ++s) {
                    const int so = tapoff + (s % 3) * 288 + (s / 3) * 16;
                    F3 a0 = ld(so), a1 = ld(so + 288);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        F3 n0, n1;
                        if (q < 3) { n0 = ld(so + (2 * q + 2) * 288); n1 = ld(so + (2 * q + 3) * 288); }
#pragma unroll
                        for (int pr = 0; pr < 6; ++pr) {
                            acc[2 * q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[PA[pr]], w[s % 3].p[PB[pr]], acc[2 * q], 0, 0, 0);
                            acc[2 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.p[PA[pr]], w[s % 3].p[PB[pr]], acc[2 * q + 1], 0, 0, 0);
                        }
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
#pragma unroll
                        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy) {
                                const int r = h - dy;
                                if (r >= 0 && r < 8) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.p[PA[pr]], w[dy].p[PB[pr]], acc[r], 0, 0, 0);
                            }
                        __builtin_amdgcn_sched_barrier(0);
                        if (h < 9) f = fn;
                    }
                }
                {
                    const int so = ((g < 3 ? g : 0) * 18 + 2) * 16 + 2 * 10 * 18 * 16;
                    F3 a0 = ld(so), a1 = ld(so + 288);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        F3 n0, n1;
                        if (q < 3) { n0 = ld(so + (2 * q + 2) * 288); n1 = ld(so + (2 * q + 3) * 288); }
#pragma unroll
                        for (int pr = 0; pr < 6; ++pr) {
                            acc[2 * q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[PA[pr]], w[0].p[PB[pr]], acc[2 * q], 0, 0, 0);
                            acc[2 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.p[PA[pr]], w[0].p[PB[pr]], acc[2 * q + 1], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (q < 3) { a0 = n0; a1 = n1; }
                    }
                }
            }
        }
        for (int r = 0; r < 8; ++r) sum += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (blockIdx.x == 17 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int VAR>
static void run(const char* name, double flop_per_iter_per_wave, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 512 * 256 * sizeof(float)); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int shm = VAR >= 4 ? 3 * 17280 : 0;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<VAR>, dim3(512), dim3(256), shm, 0, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-58s %7.3f ms  %7.1f TFLOP/s (fp32-equivalent)  clock %4.0f MHz\n", name, ms, flop_per_iter_per_wave * iters * 2048 / (ms * 1e-3) / 1e12,
           (double)h[0] / (double)h[1] * 100.0);
    hipFree(out); hipFree(clk);
}

int main() {
    const double t32 = 2.0 * 16 * 16 * 4, tb = 2.0 * 16 * 16 * 32;
    run<0>("fp32 16x16x4, random operands", 32 * t32, 40000);
    run<1>("bf16 16x16x32, random, 6 per fp32-equivalent K-step", 8 * tb, 30000);
    run<2>("bf16 32x32x16, random, 6 per fp32-equivalent K-step", 4 * 2.0 * 32 * 32 * 16, 30000);
    run<3>("bf16 16x16x32, split planes (2^-8 per plane)", 8 * tb, 30000);
    run<4>("split K loop + LDS reads, 2-row pipeline", 56 * tb, 4000);
    run<5>("split K loop + LDS reads, halo-row reuse", 56 * tb, 4000);
    return 0;
}
