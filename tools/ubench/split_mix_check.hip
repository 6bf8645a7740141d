// Bit-exactness check of the two forms of the two-term fp16 split (split_f16.h): v_cvt / v_pk_fma (compiler) against v_fma_mix{lo,hi}_f16.
// hipcc --offload-arch=gfx950 -O3 -o split_mix_check split_mix_check.hip && ./split_mix_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_ref(const float4 v, const float s, uint2& h, uint2& l) {
    const f32x2_t a = {v.x * s, v.y * s}, b = {v.z * s, v.w * s};
    const f16x2_t ha = __builtin_convertvector(a, f16x2_t), hb = __builtin_convertvector(b, f16x2_t);
    const f32x2_t ra = a - __builtin_convertvector(ha, f32x2_t), rb = b - __builtin_convertvector(hb, f32x2_t);
    const f16x2_t la = __builtin_convertvector(ra, f16x2_t), lb = __builtin_convertvector(rb, f16x2_t);
    h = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
    l = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
}
__device__ __forceinline__ void split_mix(const float4 v, const float s, uint2& h, uint2& l) {
    const f32x2_t a = {v.x * s, v.y * s}, b = {v.z * s, v.w * s};
    const f16x2_t ha = __builtin_convertvector(a, f16x2_t), hb = __builtin_convertvector(b, f16x2_t);
    const unsigned uha = __builtin_bit_cast(unsigned, ha), uhb = __builtin_bit_cast(unsigned, hb);
    unsigned la, lb;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(la) : "v"(v.x), "v"(s), "v"(uha));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(la) : "v"(v.y), "v"(s), "v"(uha));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(v.z), "v"(s), "v"(uhb));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(v.w), "v"(s), "v"(uhb));
    h = make_uint2(uha, uhb);
    l = make_uint2(la, lb);
}
__global__ void k(const float4* x, const float* sc, uint4* o_ref, uint4* o_mix, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint2 h, l;
    split_ref(x[i], sc[i], h, l); o_ref[i] = make_uint4(h.x, h.y, l.x, l.y);
    split_mix(x[i], sc[i], h, l); o_mix[i] = make_uint4(h.x, h.y, l.x, l.y);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(4 * (size_t)n), s(n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        const int e = (rand() % 60) - 30;                            // tile maximum 2^e; the scale puts it at 2^14
        s[i] = ldexpf(1.f, 14 - e);
        for (int j = 0; j < 4; ++j) {
            const int kind = rand() % 8;
            float v = ldexpf((float)rand() / RAND_MAX * 2.f - 1.f, e);
            if (kind == 0) v = ldexpf(v, -(rand() % 40));            // far below the tile maximum: l underflows fp16
            if (kind == 1) v = 0.f;
            if (kind == 2) { unsigned u = (unsigned)rand() ^ ((unsigned)rand() << 16); memcpy(&v, &u, 4); if (!(fabsf(v) < ldexpf(1.f, e))) v = ldexpf(1.f, e - 1); }
            x[4 * (size_t)i + j] = v;
        }
    }
    float4* dx; float* ds; uint4 *d0, *d1;
    hipMalloc(&dx, n * sizeof(float4)); hipMalloc(&ds, n * 4); hipMalloc(&d0, n * sizeof(uint4)); hipMalloc(&d1, n * sizeof(uint4));
    hipMemcpy(dx, x.data(), n * sizeof(float4), hipMemcpyHostToDevice); hipMemcpy(ds, s.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, d0, d1, n);
    std::vector<unsigned> a(4 * (size_t)n), b(4 * (size_t)n);
    hipMemcpy(a.data(), d0, n * sizeof(uint4), hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * sizeof(uint4), hipMemcpyDeviceToHost);
    long long bad = 0;
    for (size_t i = 0; i < a.size(); ++i) if (a[i] != b[i]) { if (bad < 5) printf("mismatch at %zu: %08x vs %08x (x = %g %g, s = %g)\n", i, a[i], b[i], x[(i / 4) * 4 + 2 * (i % 2)], x[(i / 4) * 4 + 2 * (i % 2) + 1], s[i / 4]); ++bad; }
    printf("%d quads, %lld mismatching words\n", n, bad);
    return bad != 0;
}
