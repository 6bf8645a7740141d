// Split matrix mode (da_set_matrix_mode(2)): the device-side pieces shared by the 3x3x3 kernels that run fp32 convolutions on the fp16
// matrix pipe (conv3d_mfma.hip, conv3d_s2n.hip, conv3d_up2.hip).
#pragma once
#include <hip/hip_runtime.h>

// Split mode: fp32 products on the fp16 matrix pipe.  A staged tile is scaled by a power of two s (exact) so that its largest magnitude
// lies in [2^14, 2^15), then every value is split into two fp16 terms: h = fp16(x s), l = fp16(x s - h), both round-to-nearest-even, the
// subtraction exact in fp32.  h carries 11 significand bits, l 11 of the remainder's 13: |x s - h - l| <= 2^-22 |x s| (worst case; the
// remainder loses two bits, so typically ~2^-24) for every element within 2^-18 of the tile's maximum -- below that l leaves fp16's normal
// range and the ABSOLUTE error stays at 2^-40 of the tile maximum.  A product is three MFMAs, a.h b.l + a.l b.h + a.h b.h (small terms
// first, fp32 accumulate); the dropped a.l b.l is <= 2^-22 |a b|.  Per product the bound is therefore 2^-21 + 2^-22 = 7e-7 |a b| (an fp32
// multiply-add rounds at 2^-24); over a sum of K >= ~100 products of comparable size the accumulation rounding dominates both, and there
// the split -- three roundings per 32 products instead of the fmaf chain's 32 -- is the MORE accurate of the two against double
// (tests/test_gpu_split.py: 0.4 - 0.6 x the chain's error on uniform data, equal on log-normal data where single products dominate).
// Four values at a time, packed pairwise (element 0 in the low half).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));    // eight fp16 (the A / B fragment of v_mfma_f32_16x16x32_f16)
// h: two packed multiplies + two v_cvt_pk_f16_f32.  l: v_fma_mix{lo,hi}_f16 computes fma(x, s, -h) in fp32 straight from the packed fp16 h and
// rounds it into one half of the destination -- four instructions per quad where the compiler's form (widen h, packed fma, convert) takes
// eight; the same bits (tools/ubench/split_mix_check.hip: 4M quads, l underflow and zeros included).
__device__ __forceinline__ void da_split2(const float4 v, const float s, uint2& h, uint2& l) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t a = {v.x * s, v.y * s}, b = {v.z * s, v.w * s};
    const f16x2_t ha = __builtin_convertvector(a, f16x2_t), hb = __builtin_convertvector(b, f16x2_t);      // v_cvt_pk_f16_f32 (RNE)
    const unsigned uha = __builtin_bit_cast(unsigned, ha), uhb = __builtin_bit_cast(unsigned, hb);
    unsigned la, lb;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(la) : "v"(v.x), "v"(s), "v"(uha));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(la) : "v"(v.y), "v"(s), "v"(uha));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(v.z), "v"(s), "v"(uhb));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(v.w), "v"(s), "v"(uhb));
    h = make_uint2(uha, uhb);
    l = make_uint2(la, lb);
}
// largest magnitude of a quad, folded into a running maximum (NaN operands are ignored by v_max: they still propagate through the split)
// (two v_max3_f32 with |.| source modifiers; fmaxf(fabsf()) compiles to seven instructions per quad: a canonicalising v_max per operand)
__device__ __forceinline__ float da_absmax4(float m, const float4 v) {
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(v.x), "v"(v.y), "v"(m));
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(m) : "v"(v.z), "v"(v.w), "v"(r));
    return m;
}
// wave-wide maximum of non-negative floats (their bit patterns order like integers): two quad permutes, half-row and row mirrors (DPP, VALU
// only), then the four rows through v_readlane -- the result is wave-uniform (SGPR)
__device__ __forceinline__ float da_wave_max_nonneg(float m) {
    int v = __float_as_int(m);
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true));     // row_half_mirror
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true));     // row_mirror
    const int r = max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
    return __int_as_float(r);
}
// Power-of-two scale exponent of a tile whose largest magnitude is m: m 2^e in [2^14, 2^15) (fp16 overflows at 65504), clamped to +-100;
// an all-zero (or denormal) tile gets +100, i.e. it counts as "very small" and never constrains the exponents of its neighbours.
// da_pow2(e) = 2^e for e in [-126, 127] (clamped outside: callers pass either a tile's own exponent, |e| <= 100, or go through da_acc_factor).
// Non-finite operands: an Inf makes h = Inf and l = Inf - Inf = NaN, so the outputs that touch the tile are NaN where the fp32 matrix instructions give
// +-Inf or NaN (an fp32 convolution over a tile with an Inf is NaN as soon as two taps disagree in sign); |x| > 2^114 saturates the +-100 clamp the same way.
// Non-finite in, non-finite out -- not the same non-finite.
constexpr int kSplitEmax = 100;
__device__ __forceinline__ int da_scale_exp(float m) {
    const int ef = (__float_as_int(m) >> 23) & 255;
    const int e = 141 - ef;
    return ef == 0 ? kSplitEmax : (e > kSplitEmax ? kSplitEmax : (e < -kSplitEmax ? -kSplitEmax : e));
}
__device__ __forceinline__ float da_pow2(int e) { e = e < -126 ? -126 : (e > 127 ? 127 : e); return __int_as_float((e + 127) << 23); }
// Factor that brings running sums from the unit 2^-Eacc into the unit 2^-Enew, d = Enew - Eacc.  d <= 40 by construction (every caller caps a later
// item's E at 40 above the smallest E so far); a NEGATIVE d is unbounded (|E| <= 200) and one fp32 factor only reaches 2^-126.  Below that the old
// sums (|sum| < 2^42 in their unit: K <= 2^12 products of two fp16 magnitudes < 2^15) are < 2^-84 in the new unit, whose own products are staged
// with their largest magnitudes at 2^11..2^15 each: 2^-100 of what one fp32 rounding of the new sum discards.  The factor is then 0, not a clamped 2^-126
// (which would leave the old sums wrong by a power of two).
__device__ __forceinline__ float da_acc_factor(int d) { return d < -126 ? 0.f : da_pow2(d); }
// A workgroup's (4 waves) largest magnitude of a staged tile: every wave publishes its maximum (one float per wave at `slot`, 16 bytes of LDS),
// a barrier, everyone reads the four.  The barrier doubles as "all waves are done with the tile in LDS" wherever the caller needs that.
__device__ __forceinline__ float da_block_max4(float m, float* slot, int wave, int lane) {
    m = da_wave_max_nonneg(m);
    if (lane == 0) slot[wave] = m;
    __syncthreads();
    const float4 mm = *reinterpret_cast<const float4*>(slot);
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fmaxf(fmaxf(mm.x, mm.y), fmaxf(mm.z, mm.w)))));
}
