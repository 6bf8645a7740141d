// Staging helpers, tile walk and the launch parameter block shared by the stride-1 3x3x3 implicit-GEMM kernels (conv3d_mfma.hip: every matrix
// mode; conv3d_fwdsp.hip: the split mode's forward / data-gradient kernel with the weights in LDS).  Everything device-side lives in an anonymous
// namespace (one copy per translation unit); the two plain structs that cross translation units have external names.
#pragma once
#include "common.h"
#include "split_f16.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));      // four bf16 (the A / B fragment of v_mfma_f32_16x16x16_bf16)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));    // eight bf16 (the A / B fragment of v_mfma_f32_16x16x32_bf16)
#include <type_traits>

struct DaC3S2dSrc { int cin, D0, H0, W0; };
struct DaC3FwdP {
    const float* in1; const float* in2; int C1, C2;
    const float* wp; const float* bias;
    float* out1; float* out2; int Cs1, Cs2;
    int N, D, H, W, Cout, NT, ntz, nty, ntx, ntiles, nblocks;
    float slope;
    unsigned masks[16]; int maskmode;   // tap masks (stride-2 via space-to-depth): 0 none, 1 per channel chunk, 2 per blockIdx.y
    double* stats_partial;              // optional [gridDim.x][2][Cout]: per-workgroup sum / sum of squares of the (pre-activation) output
    unsigned long long* clk;
    const float* ps1; const float* pt1; const float* ps2; const float* pt2; float pslope1, pslope2;   // PRO: per-channel scale / shift / act slope still to be applied to in1 / in2
    int* dyn_ctr;    // DYN: tile counters [gridDim.y][8 XCDs], zeroed by the pack kernel of the same call
    const int4* tiles;   // (n, z0, y0, x0) of every tile in brick order, written by the pack kernel of the same call: the persistent loop
                         // reads one entry per item through the scalar cache instead of decomposing the position (~10 integer divisions)
    int prio_ranks;  // co-resident workgroups per CU taking turns at the top wave priority (0: off)
    DaC3S2dSrc s2in;     // MASKED forward: in1 is the ORIGINAL tensor of a stride-2 layer, read as its space-to-depth view (cin > 0)
    const int* wexp; // SP: power-of-two exponent of every channel chunk of the packed weights (pack_split_weights_kernel)
    DaC3S2dSrc s2out;    // MASKED data gradient: the 8 * cin output channels are scattered to the original-resolution gradient (cin > 0)
    int ablate;      // diagnostic only (env DA_ABLATE): 1 no staging loads (offsets forced out of range), 2 no epilogue, 4 no LDS writes + barriers
    // STATS == 2 (a data gradient whose output dx is the gradient with respect to act(BN(y))): y = that layer's raw conv output (same shape as out1),
    // bst_par = its statistics rows [mean | rstd | scale | shift][Cs1]; stats_partial then receives (sum dz, sum dz (y - mean)), dz = dx act'(y scale + shift)
    const float* bst_y; const float* bst_par; float bst_slope;
};
// conv3d_fwdsp.hip: split mode, fp32 tensors, 8-channel chunks.  nrep: N-tiles per workgroup (1 | 2); stats: 0 none, 1 BatchNorm sums of the output,
// 2 BatchNorm-backward sums (data gradient); pro: input prologue; pair: paired staging of chunk pairs (nrep 1, no prologue, stats != 2)
bool da_conv3_fwdsp_enabled();
int da_conv3_fwdsp_launch(const DaC3FwdP& p, int gy, int nrep, int stats, int pro, int pair, hipStream_t st);

namespace {

constexpr int TY = 8, TX = 16, HY = TY + 2, HX = TX + 2;

// bijective XCD-aware remap: consecutive tiles land on the same XCD (shared halos hit that XCD's L2)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, loc = bid / 8;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

// Staging of an (HZ x HY x HX) halo tile of CK channels into LDS as [voxel][CK]; out-of-volume voxels are zero
// (padding 1).  Split T14-style: stage_load issues the global loads into registers (one work item AHEAD, so their
// latency hides under the current item's MFMAs), stage_write drops them into LDS after the barrier that retires the
// previous tile.  The LDS image is linear in the flat (voxel, channel-quad) index, so the write is one ds_write_b128.
template <int CK, int HZ> struct StageGeom {
    static constexpr int Q = CK / 4;
    static constexpr int TOTAL = HZ * HY * HX * Q;
    static constexpr int NIT = (TOTAL + 255) / 256;
};

// Global loads go through a buffer descriptor over ONE sample's tensor: an out-of-volume voxel gets byte offset
// 0xFFFFFFFF, which the hardware range check turns into zeros -- no exec-mask branch per load, so the NIT loads issue
// back to back.  The (hz, hy, hx) decomposition of the flat index is done once and then stepped by 256/Q voxels per
// iteration with carries (two compares) instead of two magic-number divisions per iteration.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t da_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ unsigned da_bf16x2(float lo, float hi) {      // round-to-nearest-even, ONE v_cvt_pk_bf16_f32 for the pair
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float4 da_buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
// HB: the tensor is stored as bf16 (bf16 activation storage, common.h): a channel quad is 8 bytes and is widened to fp32 on arrival, so
// everything behind the load (prologue arithmetic, the conversion into the bf16 LDS image -- exact for these values) is shared with the
// fp32-storage kernels.  Only instantiated for the bf16 matrix mode (BF && !SP).
template <bool HB> struct HbEl { static constexpr unsigned ES = HB ? 2u : 4u; };
// RAW (bf16 storage, no arithmetic between the load and the bf16 LDS image): the eight bytes travel untouched in .x / .y -- widening them to
// fp32 only to round them back was ~7 VALU instructions per staged quad in a kernel whose busiest pipe is the VALU.
template <bool HB, bool RAW = false> __device__ __forceinline__ float4 da_buf_loadq(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    if constexpr (HB) {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t u = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
        if constexpr (RAW) return make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), 0.f, 0.f);
        else return da_unpack_bf16x4(make_uint2(u[0], u[1]));
    } else return da_buf_load4(r, byte_off);
}
// buffer descriptor over sample n of a tensor with `sample` elements per sample
template <bool HB> __device__ __forceinline__ __amdgpu_buffer_rsrc_t da_rsrc_n(const float* base, long long n, long long sample) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(base) + n * sample * (long long)HbEl<HB>::ES), 0, (unsigned)(sample * HbEl<HB>::ES), 0x00020000);
}

template <int CK, int HZ, int IT0 = 0, int IT1 = StageGeom<CK, HZ>::NIT, bool HB = false, bool RAW = false>
__device__ __forceinline__ void stage_load(float4* pre, const float* __restrict__ src, int Cs, int choff,
                                           int n, int z0, int y0, int x0, int D, int H, int W, unsigned* vmask = nullptr) {
    constexpr int Q = StageGeom<CK, HZ>::Q, TOTAL = StageGeom<CK, HZ>::TOTAL;
    constexpr int STEP = 256 / Q;                        // voxels per iteration
    constexpr int SX = STEP % HX, SY = (STEP / HX) % HY, SZ = STEP / (HX * HY);
    const long long sample = (long long)D * H * W * Cs;
    const __amdgpu_buffer_rsrc_t rs = da_rsrc_n<HB>(src, n, sample);
    int idx = threadIdx.x + IT0 * 256;
    asm volatile("" : "+v"(idx));                       // keep the decomposition out of the persistent loop's invariants
    const int c4 = idx % Q; int hv = idx / Q;
    int hx = hv % HX; int t = hv / HX;
    int hy = t % HY; int hz = t / HY;
    const int cofs = choff + c4 * 4;
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
        const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool inb = (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && (hv < TOTAL / Q);
        const unsigned off = ((unsigned)((z * H + y) * W + x) * (unsigned)(Cs) + (unsigned)(cofs)) * (unsigned)(HbEl<HB>::ES);
        pre[it - IT0] = da_buf_loadq<HB, RAW>(rs, inb ? off : 0xFFFFFFFFu);
        if (vmask) *vmask |= (inb ? 1u : 0u) << (it - IT0);
        hv += STEP;
        hx += SX; const int cx = hx >= HX ? 1 : 0; hx -= cx * HX;
        hy += SY + cx; const int cy = hy >= HY ? 1 : 0; hy -= cy * HY;
        hz += SZ + cy;
    }
}

// The same staging for a tensor that is only VIRTUALLY space-to-depth (stride-2 layers, conv3d_s2.hip): (z, y, x) and `choff` address
// S[q][r * cin + c] = X[2 q + r][c] (r = (rz, ry, rx) parity), and the loads go straight to X (cin channels, D0 x H0 x W0) -- the
// space_to_depth2 copy pass and its 8 * cin-channel tensor disappear.  A chunk lies inside one parity (cin % CK == 0).
using S2dSrc = DaC3S2dSrc;
template <int CK, int HZ, int IT0 = 0, int IT1 = StageGeom<CK, HZ>::NIT, bool HB = false>
__device__ __forceinline__ void stage_load_s2d(float4* pre, const float* __restrict__ src, S2dSrc s2, int choff,
                                               int n, int z0, int y0, int x0, int D, int H, int W) {
    constexpr int Q = StageGeom<CK, HZ>::Q, TOTAL = StageGeom<CK, HZ>::TOTAL;
    constexpr int STEP = 256 / Q;
    constexpr int SX = STEP % HX, SY = (STEP / HX) % HY, SZ = STEP / (HX * HY);
    const long long sample = (long long)s2.D0 * s2.H0 * s2.W0 * s2.cin;
    const __amdgpu_buffer_rsrc_t rs = da_rsrc_n<HB>(src, n, sample);
    int idx = threadIdx.x + IT0 * 256;
    asm volatile("" : "+v"(idx));
    const int c4 = idx % Q; int hv = idx / Q;
    int hx = hv % HX; int t = hv / HX;
    int hy = t % HY; int hz = t / HY;
    const int r = choff / s2.cin;                                  // wave-uniform: the chunk's parity
    const int cofs = choff - r * s2.cin + c4 * 4;
    const int rz = (r >> 2) & 1, ry = (r >> 1) & 1, rx = r & 1;
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
        const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
        const int zs = 2 * z + rz, ys = 2 * y + ry, xs = 2 * x + rx;
        const bool inb = (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && (hv < TOTAL / Q)
                         && zs < s2.D0 && ys < s2.H0 && xs < s2.W0;
        const unsigned off = ((unsigned)((zs * s2.H0 + ys) * s2.W0 + xs) * (unsigned)(s2.cin) + (unsigned)(cofs)) * (unsigned)(HbEl<HB>::ES);
        pre[it - IT0] = da_buf_loadq<HB>(rs, inb ? off : 0xFFFFFFFFu);
        hv += STEP;
        hx += SX; const int cx = hx >= HX ? 1 : 0; hx -= cx * HX;
        hy += SY + cx; const int cy = hy >= HY ? 1 : 0; hy -= cy * HY;
        hz += SZ + cy;
    }
}

// LeakyReLU / ReLU / identity as max(z, z * s) with s = slope in [0, 1) or s = 1 for "no activation": two VALU operations, no
// compares, and bit-identical to da_act() for every finite or non-finite z (z > 0: z; z < 0: z * s >= z; -0 and NaN propagate alike).
__device__ __forceinline__ float da_act01(float z, float s) { return fmaxf(z, z * s); }

// Input prologue (PRO variants): the staged tensor is a RAW convolution output whose BatchNorm + LeakyReLU has not been applied
// yet; it is applied here, on the way into LDS, with exactly the expression of bn_act_fwd_kernel (norm_act.hip) so the result is
// bit-identical to materialising the activated tensor first.  Padding (out-of-volume voxels, mask bit clear) stays zero.
template <int CK, int HZ, int IT0, int IT1, bool BF, int ZPAD = 0>     // ZPAD: quads of padding after every z plane of the LDS image (bank spreading)
__device__ __forceinline__ void stage_write_pro(float* __restrict__ lds, const float4* pre, unsigned vmask, float4 sc, float4 sf, float slope) {
    constexpr int TOTAL0 = StageGeom<CK, HZ>::TOTAL;
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
        const int idx0 = threadIdx.x + it * 256;
        const int idx = ZPAD ? idx0 + ZPAD * (idx0 / (HY * HX * StageGeom<CK, HZ>::Q)) : idx0;
        if (idx0 < TOTAL0) {
            const float4 t = pre[it - IT0];
            const bool ok = ((vmask >> (it - IT0)) & 1u) != 0;
            float4 v;
            v.x = ok ? da_act01(t.x * sc.x + sf.x, slope) : 0.f; v.y = ok ? da_act01(t.y * sc.y + sf.y, slope) : 0.f;
            v.z = ok ? da_act01(t.z * sc.z + sf.z, slope) : 0.f; v.w = ok ? da_act01(t.w * sc.w + sf.w, slope) : 0.f;
            if constexpr (BF) reinterpret_cast<uint2*>(lds)[idx] = make_uint2(da_bf16x2(v.x, v.y), da_bf16x2(v.z, v.w));
            else reinterpret_cast<float4*>(lds)[idx] = v;
        }
    }
}
// the same prologue applied IN PLACE to the parked quads (split mode: the tile's largest magnitude must be known before anything is written)
template <int IT0, int IT1>
__device__ __forceinline__ void stage_pro_apply(float4* pre, unsigned vmask, float4 sc, float4 sf, float slope) {
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
        const float4 t = pre[it - IT0];
        const bool ok = ((vmask >> (it - IT0)) & 1u) != 0;
        pre[it - IT0].x = ok ? da_act01(t.x * sc.x + sf.x, slope) : 0.f; pre[it - IT0].y = ok ? da_act01(t.y * sc.y + sf.y, slope) : 0.f;
        pre[it - IT0].z = ok ? da_act01(t.z * sc.z + sf.z, slope) : 0.f; pre[it - IT0].w = ok ? da_act01(t.w * sc.w + sf.w, slope) : 0.f;
    }
}

// BF: the LDS image holds bf16 (same [voxel][CK] order, 8 bytes per channel quad): converted once here instead of at every tap
// SP: two fp16 planes (h, l of da_split2 at the tile's scale `sps`), each in the BF layout, TOTAL quads apart
template <int CK, int HZ, int IT0 = 0, int IT1 = StageGeom<CK, HZ>::NIT, bool BF = false, bool SP = false, int ZPAD = 0, bool RAW = false>
__device__ __forceinline__ void stage_write(float* __restrict__ lds, const float4* pre, const float sps = 1.f) {
    static_assert(!RAW || (BF && !SP), "raw staging: bf16 storage into the one-plane bf16 image");
    constexpr int TOTAL0 = StageGeom<CK, HZ>::TOTAL, TOTAL = TOTAL0 + HZ * ZPAD;
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
        const int idx0 = threadIdx.x + it * 256;
        const int idx = ZPAD ? idx0 + ZPAD * (idx0 / (HY * HX * StageGeom<CK, HZ>::Q)) : idx0;
        if (idx0 < TOTAL0) {
            if constexpr (SP) {
                uint2 h, l; da_split2(pre[it - IT0], sps, h, l);
                reinterpret_cast<uint2*>(lds)[idx] = h; reinterpret_cast<uint2*>(lds)[idx + TOTAL] = l;
            } else if constexpr (RAW) {
                reinterpret_cast<uint2*>(lds)[idx] = make_uint2(__float_as_uint(pre[it - IT0].x), __float_as_uint(pre[it - IT0].y));
            } else if constexpr (BF) {
                const float4 v = pre[it - IT0];
                reinterpret_cast<uint2*>(lds)[idx] = make_uint2(da_bf16x2(v.x, v.y), da_bf16x2(v.z, v.w));
            } else reinterpret_cast<float4*>(lds)[idx] = pre[it - IT0];
        }
    }
}
// largest magnitude among the quads a thread parks for one tile (iterations past the tile hold zeros: their offsets were out of range)
template <int NITS>
__device__ __forceinline__ float stage_absmax(const float4* pre) {
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < NITS; ++it) m = da_absmax4(m, pre[it]);
    return m;
}

// The same staging loads, one at a time: a cursor that carries the incremental (hz, hy, hx) decomposition so the loads of the
// NEXT work item can be spread over the K-steps of the current one.  `valid` = false turns every offset out of range (zeros, no
// memory traffic), so the loads are issued on every item without a branch around them and hipcc's vmcnt bookkeeping stays exact.
template <int CK, int HZ, bool HB = false, bool RAW = false> struct StageCursor {
    __amdgpu_buffer_rsrc_t rs;
    int hv, hx, hy, hz, cofs;
    int z0, y0, x0, D, H, W, Cs;
    bool valid, last_inb;
    __device__ __forceinline__ void init(const float* __restrict__ src, int Cs_, int choff, int n, int z0_, int y0_, int x0_,
                                         int D_, int H_, int W_, bool valid_) {
        constexpr int Q = StageGeom<CK, HZ>::Q;
        const long long sample = (long long)D_ * H_ * W_ * Cs_;
        rs = da_rsrc_n<HB>(src, n, sample);
        int idx = threadIdx.x;
        asm volatile("" : "+v"(idx));
        const int c4 = idx % Q; hv = idx / Q;
        hx = hv % HX; const int t = hv / HX;
        hy = t % HY; hz = t / HY;
        cofs = choff + c4 * 4;
        z0 = z0_; y0 = y0_; x0 = x0_; D = D_; H = H_; W = W_; Cs = Cs_; valid = valid_;
    }
    __device__ __forceinline__ float4 next() {
        constexpr int Q = StageGeom<CK, HZ>::Q, TOTAL = StageGeom<CK, HZ>::TOTAL;
        constexpr int STEP = 256 / Q;
        constexpr int SX = STEP % HX, SY = (STEP / HX) % HY, SZ = STEP / (HX * HY);
        const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool inb = valid && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && (hv < TOTAL / Q);
        const unsigned off = ((unsigned)((z * H + y) * W + x) * (unsigned)(Cs) + (unsigned)(cofs)) * (unsigned)(HbEl<HB>::ES);
        const float4 v = da_buf_loadq<HB, RAW>(rs, inb ? off : 0xFFFFFFFFu);
        last_inb = inb;
        hv += STEP;
        hx += SX; const int cx = hx >= HX ? 1 : 0; hx -= cx * HX;
        hy += SY + cx; const int cy = hy >= HY ? 1 : 0; hy -= cy * HY;
        hz += SZ + cy;
        return v;
    }
};
// Per-thread constants of the halo staging (replaces the cursor inside the persistent loop): iteration `it` of this thread covers
// halo voxel (hz, hy, hx) = pk[it] >> 16, (pk[it] >> 8) & 255, pk[it] & 255 and channel quad threadIdx.x % Q.  The byte offsets of one
// item's NIT loads are computed in one go at the top of the item: 7 VALU operations per load for a tile whose whole halo lies
// inside the volume (the common case, a wave-uniform branch around pure arithmetic), bounds checks only on boundary tiles.
template <int CK, int HZ, bool VO = true> struct StageMap {      // VO = false: no vo[] (kernels without NIT registers to spare): the interior offsets are recomputed from pk[]
    static constexpr int NIT = StageGeom<CK, HZ>::NIT, Q = StageGeom<CK, HZ>::Q, TOTAL = StageGeom<CK, HZ>::TOTAL;
    unsigned pk[NIT];
    int vo[VO ? NIT : 1]; // voxel index of the halo voxel relative to the halo's corner, (hz H + hy) W + hx: constant for the whole launch
    int c4x4;
    bool small;           // 6 H W < 2^24: the offsets of an interior tile are one v_mad_u32_u24 per load
    __device__ __forceinline__ void init(int H, int W) {
        c4x4 = ((int)threadIdx.x % Q) * 4;
        small = (long long)HZ * H * W < (1ll << 24);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int hv = ((int)threadIdx.x + it * 256) / Q;
            const int hx = hv % HX, t = hv / HX, hy = t % HY, hz = t / HY;
            pk[it] = (hv < TOTAL / Q) ? ((unsigned)hz << 16 | (unsigned)hy << 8 | (unsigned)hx) : 0xFFFF0000u;     // past the tile: hz = 65535 is never inside
            if constexpr (VO) vo[it] = (hv < TOTAL / Q) ? (hz * H + hy) * W + hx : 0;
        }
    }
    // tile-level part (wave-uniform): is the whole halo inside the volume; per-thread base (bytes) of an interior tile's offsets
    struct Tile { bool interior, valid; int z0, y0, x0, H, W, D, Cs4, cofs4; unsigned base; };
    __device__ __forceinline__ Tile tile(int z0, int y0, int x0, int D, int H, int W, int Cs, int choff, bool valid, int es = 4) const {      // es: bytes per stored element
        Tile t;
        t.valid = valid; t.z0 = z0; t.y0 = y0; t.x0 = x0; t.D = D; t.H = H; t.W = W;
        t.interior = small && valid && z0 >= 1 && z0 + HZ - 2 < D && y0 >= 1 && y0 + HY - 2 < H && x0 >= 1 && x0 + HX - 2 < W;
        t.Cs4 = Cs * es; t.cofs4 = (choff + c4x4) * es;
        t.base = (unsigned)(((z0 - 1) * H + (y0 - 1)) * W + (x0 - 1)) * (unsigned)t.Cs4 + (unsigned)t.cofs4;
        return t;
    }
    __device__ __forceinline__ unsigned offset(const Tile& t, int it) const {
        const int hz = (int)(pk[it] >> 16), hy = (int)((pk[it] >> 8) & 255u), hx = (int)(pk[it] & 255u);
        if (t.interior) {
            unsigned o;
            if constexpr (VO) o = __umul24((unsigned)vo[it], (unsigned)t.Cs4) + t.base;
            else o = (unsigned)((hz * t.H + hy) * t.W + hx) * (unsigned)t.Cs4 + t.base;
            return ((it + 1) * 256 <= TOTAL || hz != 0xFFFF) ? o : 0xFFFFFFFFu;
        }
        const int z = t.z0 - 1 + hz, y = t.y0 - 1 + hy, x = t.x0 - 1 + hx;
        const bool inb = t.valid && (unsigned)z < (unsigned)t.D && (unsigned)y < (unsigned)t.H && (unsigned)x < (unsigned)t.W;
        return inb ? (unsigned)(((z * t.H + y) * t.W + x) * t.Cs4 + t.cofs4) : 0xFFFFFFFFu;
    }
};
__device__ __forceinline__ void da_buf_store4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r, byte_off, 0, 0);
}
template <bool HB> __device__ __forceinline__ void da_buf_storeq(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    if constexpr (HB) {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t u = {da_pack_bf16x2(v[0], v[1]), da_pack_bf16x2(v[2], v[3])};
        __builtin_amdgcn_raw_buffer_store_b64(u, r, byte_off, 0, 0);
    } else da_buf_store4(r, byte_off, v);
}
template <bool B> struct BoolC { static constexpr bool value = B; };
template <int V> struct IntC { static constexpr int value = V; };
// Wave priority rotation (experiment, DA_PRIO_ROT=1; off by default).  The persistent kernels keep 2 - 3 workgroups per CU alive for the
// whole launch, and the CU arbitrates instruction issue between their waves by priority, then by AGE: with equal priorities the
// first-dispatched workgroup of a CU runs ~20 % faster than the last one for the whole kernel (DA_CLK=1 DA_CLK_DUMP=1: lifetimes 1.39 /
// 1.67 / 1.91 ms on every CU for equal work, split-mode 48 -> 16 forward).  Rotating s_setprio over the co-resident workgroups once per work
// item (rank = dispatch round, 256 workgroups per round) halves the spread of the finish times (508 -> 253 us) and changes the kernel time
// by nothing (2.27 -> 2.28 ms): the kernel is power-bound, the early finishers' CUs were not wasted -- the survivors ran at a higher clock.
__device__ __forceinline__ void da_setprio(int p) {
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}
// lane ^ 1 / lane ^ 2 exchanges inside a lane quad as DPP quad_perm moves (VALU, no trip through the LDS crossbar like __shfl_xor)
__device__ __forceinline__ float da_quad_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float da_quad_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]


// ---------------------------------------------------------------------------------------------------
// XCD-aware tile walk.  Workgroup b lands on XCD b % 8 (round-robin dispatch) and each XCD has its own 4 MiB L2, so the tiles
// that are resident at the same time on one XCD should be spatial neighbours: the halo voxels they share are then fetched from
// HBM once instead of once per tile.  Tiles are therefore ordered brick by brick (BX x BY x BZ tiles, ragged at the volume
// edge), every XCD owns one contiguous eighth of that order, and its J workgroups walk it interleaved (step k: positions
// lo + k J + j), i.e. at any moment an XCD works on ~J consecutive positions = one or two bricks.
// ---------------------------------------------------------------------------------------------------
struct TileWalk { int lo, J, cnt; };

__device__ __forceinline__ TileWalk tile_walk(int ntiles, int G = gridDim.x, int b = blockIdx.x) {
    const int X = (G % 8 == 0) ? 8 : 1;
    const int J = G / X, xcd = b % X, j = b / X;
    const int lo = (int)((long long)ntiles * xcd / X), hi = (int)((long long)ntiles * (xcd + 1) / X);
    TileWalk w;
    w.lo = lo + j; w.J = J;
    w.cnt = (hi - lo > j) ? (hi - lo - j + J - 1) / J : 0;
    return w;
}

template <int BX, int BY, int BZ>
__device__ __forceinline__ void brick_tile(int pos, int ntx, int nty, int ntz, int& n, int& tx, int& ty, int& tz) {
    const int per_sample = ntx * nty * ntz;
    n = pos / per_sample; int r = pos - n * per_sample;
    const int zb = r / (BZ * nty * ntx); r -= zb * (BZ * nty * ntx);
    const int sz = min(BZ, ntz - zb * BZ);
    const int yb = r / (sz * BY * ntx); r -= yb * (sz * BY * ntx);
    const int sy = min(BY, nty - yb * BY);
    const int xb = r / (sz * sy * BX); r -= xb * (sz * sy * BX);
    const int sx = min(BX, ntx - xb * BX);
    const int lx = r % sx; r /= sx;
    const int ly = r % sy; const int lz = r / sy;
    tx = xb * BX + lx; ty = yb * BY + ly; tz = zb * BZ + lz;
}

using FwdP = DaC3FwdP;

}  // namespace
