// What the K loop of the two-term fp16 split can sustain (random operands, power-limited clock), by how its A fragments are fed:
//   0  v_mfma_f32_16x16x32_f16 only, 8 accumulators, 3 products per fp32-equivalent K-step
//   1  + the two planes' ds_read_b128 per (row, K-step) of the conv kernel's LDS tile, row PAIRS (2 accumulators in flight per product)
//   2  the same with row blocks of four (4 accumulators in flight per product)
//   3  halo-row reuse: lane group = (dz, dx) combination, one fragment per halo row serves dy = 0, 1, 2 (half the LDS reads)
//   4  variant 1 with TWO N-tiles (6 MFMAs per fragment pair)
// Every kernel runs ~10 ms; clock = shader cycles (s_memtime) / wall (s_memrealtime, 100 MHz) of one workgroup.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/split_f16_kloop.hip -o tools/ubench/split_f16_kloop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned rng(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ unsigned short rand_h(unsigned& s, int shift_exp) {      // random fp16 in [1, 2) * 2^(14 - shift_exp), random sign
    const unsigned r = rng(s) >> 8;
    return (unsigned short)(((r >> 10) & 1u) << 15 | ((unsigned)(15 + 14 - shift_exp) << 10) | (r & 1023u));
}
__device__ __forceinline__ f16x8 rand_frag(unsigned& s, int shift_exp) {
    f16x8 f;
    for (int e = 0; e < 8; ++e) f[e] = __builtin_bit_cast(_Float16, rand_h(s, shift_exp));
    return f;
}
struct F2 { f16x8 p[2]; };
constexpr int PLANE = 6 * 10 * 18 * 16;     // bytes
template <int VAR>
__global__ void __launch_bounds__(256, 2) k(float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    unsigned seed = blockIdx.x * 256 + threadIdx.x + 12345u;
    if (VAR >= 1) {
        unsigned short* l16 = reinterpret_cast<unsigned short*>(lds);
        for (int t = threadIdx.x; t < 2 * PLANE / 2; t += 256) l16[t] = rand_h(seed, 11 * (t / (PLANE / 2)));
        __syncthreads();
    }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
    constexpr int NR = VAR == 4 ? 2 : 1;
    f32x4 acc[8][NR];
    for (int r = 0; r < 8; ++r) for (int n = 0; n < NR; ++n) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const char* base = reinterpret_cast<const char*>(lds) + ((wave * 10) * 18 + i) * 16;
    auto ld = [&](int off) -> F2 { F2 f; const char* a = base + off; f.p[0] = *reinterpret_cast<const f16x8*>(a); f.p[1] = *reinterpret_cast<const f16x8*>(a + PLANE); return f; };
    if constexpr (VAR == 0) {
        F2 a[8], b;
        for (int r = 0; r < 8; ++r) for (int pl = 0; pl < 2; ++pl) a[r].p[pl] = rand_frag(seed, 11 * pl);
        for (int pl = 0; pl < 2; ++pl) b.p[pl] = rand_frag(seed, 11 * pl);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.p[PB[pr]], a[r].p[PA[pr]], acc[r][0], 0, 0, 0);
        }
    } else if constexpr (VAR == 1 || VAR == 2 || VAR == 4) {
        constexpr int RPB = VAR == 2 ? 4 : 2;
        F2 w[3][NR];
        for (int d = 0; d < 3; ++d) for (int n = 0; n < NR; ++n) for (int pl = 0; pl < 2; ++pl) w[d][n].p[pl] = rand_frag(seed, 11 * pl);
        const int tapoff = ((g >> 1) * 10 * 18 + (g & 1)) * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                const int so = tapoff + (s % 3) * 288 + (s / 3) * 16;
                F2 a[RPB];
#pragma unroll
                for (int rr = 0; rr < RPB; ++rr) a[rr] = ld(so + rr * 288);
#pragma unroll
                for (int q = 0; q < 8 / RPB; ++q) {
                    F2 n[RPB];
                    if (q + 1 < 8 / RPB) {
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr) n[rr] = ld(so + (RPB * q + RPB + rr) * 288);
                    }
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int nn = 0; nn < NR; ++nn)
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr)
                            acc[RPB * q + rr][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[s % 3][nn].p[PB[pr]], a[rr].p[PA[pr]], acc[RPB * q + rr][nn], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q + 1 < 8 / RPB) {
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr) a[rr] = n[rr];
                    }
                }
            }
        }
    } else {
        F2 w[3];
        for (int d = 0; d < 3; ++d) for (int pl = 0; pl < 2; ++pl) w[d].p[pl] = rand_frag(seed, 11 * pl);
        const int tapoff = ((g >> 1) * 10 * 18 + (g & 1)) * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int so = tapoff + c * 32;
                F2 f = ld(so);
#pragma unroll
                for (int h = 0; h < 10; ++h) {
                    F2 fn; if (h < 9) fn = ld(so + (h + 1) * 288);
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int r = h - dy;
                            if (r >= 0 && r < 8) acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[dy].p[PB[pr]], f.p[PA[pr]], acc[r][0], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    if (h < 9) f = fn;
                }
            }
            {
                const int so = ((g < 3 ? g : 0) * 18 + 2) * 16 + 2 * 10 * 18 * 16;
                F2 a0 = ld(so), a1 = ld(so + 288);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    F2 n0, n1;
                    if (q < 3) { n0 = ld(so + (2 * q + 2) * 288); n1 = ld(so + (2 * q + 3) * 288); }
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr) {
                        acc[2 * q][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0].p[PB[pr]], a0.p[PA[pr]], acc[2 * q][0], 0, 0, 0);
                        acc[2 * q + 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0].p[PB[pr]], a1.p[PA[pr]], acc[2 * q + 1][0], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (q < 3) { a0 = n0; a1 = n1; }
                }
            }
        }
    }
    for (int r = 0; r < 8; ++r) for (int n = 0; n < NR; ++n) sum += acc[r][n][0] + acc[r][n][1] + acc[r][n][2] + acc[r][n][3];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (blockIdx.x == 17 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int VAR>
static void run(const char* name, double mfma_per_iter_per_wave, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 512 * 256 * sizeof(float)); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int shm = VAR >= 1 ? 2 * PLANE : 0;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<VAR>, dim3(512), dim3(256), shm, 0, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double raw = mfma_per_iter_per_wave * 2.0 * 16 * 16 * 32 * iters * 2048 / (ms * 1e-3) / 1e12;
    const double mhz = (double)h[0] / (double)h[1] * 100.0;
    printf("%-62s %7.3f ms  %7.1f TFLOP/s raw = %6.1f fp32-equivalent (3 products)  clock %4.0f MHz  matrix pipe busy %.2f\n", name, ms, raw, raw / 3.0, mhz,
           mfma_per_iter_per_wave * iters * 2.0 * 16.0 / (ms * 1e-3 * mhz * 1e6));
    hipFree(out); hipFree(clk);
}
int main() {
    run<0>("f16 16x16x32 only, 8 accumulators", 168, 6000);
    run<1>("K loop + LDS reads, row pairs (one N-tile)", 168, 6000);
    run<2>("K loop + LDS reads, row blocks of four (one N-tile)", 168, 6000);
    run<3>("K loop + LDS reads, halo-row reuse (one N-tile)", 168, 6000);
    run<4>("K loop + LDS reads, row pairs, two N-tiles", 336, 3000);
    return 0;
}
