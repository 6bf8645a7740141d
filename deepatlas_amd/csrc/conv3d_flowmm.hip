// The registration net's flow convolution (3x3x3, 8 + 16 -> 3 channels, full resolution; voxel_morph.py:57,82) and its data gradient on the
// matrix cores, split matrix mode (two-term fp16 split, split_f16.h).  The implicit-GEMM kernels of conv3d_mfma.hip want >= 8 output
// channels (an N-tile that is 13/16 padding costs more than the VALU kernel it replaces); here the GEMMs are shaped around the 3 channels:
//
//  forward    N = (dy, cout): 3 x 3 (+ 1 pad) = 12 of 16 columns.  P[y'][(dy, co)] = sum over the nine (dz, dx) taps and the input channels
//             of x[z + dz - 1][y'][x + dx - 1][ci] W[dz][dy][dx][ci][co] is ONE MFMA accumulation per halo row y' (K = 9 x Cin instead of
//             27 x Cin), and out[y][co] = P[y][(0, co)] + P[y + 1][(1, co)] + P[y + 2][(2, co)] is a sum over three lane groups of the same
//             wave (the weights are the MFMA's A operand, so a lane holds one voxel and the four columns 4 g .. 4 g + 3 = (dy = g, co)).
//             Six halo rows per wave give four output rows: 1.5 x the row count at a third of the K extent.
//  data grad  M = voxels, N = input channels (24 -> two N-tiles), K = (tap, cout padded to 4) = 108 of 128: the dY halo tile is 11 KB.
//
// Both are HBM-bound by design (forward: 472 MB in, 59 MB out at 160 x 192 x 160); the tiles' scales follow conv3d_mfma.hip "SP": the staged
// activation / gradient tile at its own power of two, the weights at one scale per tensor, nothing accumulates across tiles.
#include "common.h"
#include "conv3d_internal.h"
#include "split_f16.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TZ = 2, TY = 8, TX = 16;                     // output tile
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;       // halo 4 x 10 x 18
constexpr int HV = HZ * HY * HX;                           // 720 voxels

__device__ __forceinline__ __amdgpu_buffer_rsrc_t fm_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 fm_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float fm_load1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void fm_store4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r, off, 0, 0);
}
__device__ __forceinline__ void fm_tile(int t, int ntz, int nty, int ntx, int& n, int& z0, int& y0, int& x0) {      // z fastest: neighbours in z / y share the most halo
    const int tz = t % ntz; t /= ntz;
    const int ty = t % nty; t /= nty;
    const int tx = t % ntx; n = t / ntx;
    z0 = tz * TZ; y0 = ty * TY; x0 = tx * TX;
}
// largest |w| of a small weight tensor (n floats) by one 256-thread workgroup
__device__ __forceinline__ float fm_tensor_absmax(const float* __restrict__ w, int n, float* red) {
    float m = 0.f;
    for (int k = threadIdx.x; k < n; k += 256) m = fmaxf(m, fabsf(w[k]));
    return da_block_max4(m, red, (int)threadIdx.x >> 6, (int)threadIdx.x & 63);
}
// products of a split multiply, small terms first, weights as the MFMA's A operand: (w plane, x plane) = (h, l) (l, h) (h, h)
__device__ __forceinline__ f32x4 fm_mma3(f32x4 c, const f16x8 (&w)[2], const f16x8 (&x)[2]) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], x[0], c, 0, 0, 0);
    return c;
}

// ------------------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------------------
struct FwdP {
    const float* in1; const float* in2; const unsigned char* wp; const int* wexp; const float* bias; float* out;
    int N, D, H, W, Cout, ntz, nty, ntx, ntiles;
    float slope;
};

// packed A operand (weights): [step][plane][lane][8 fp16]; lane (n, g): K block b = 4 step + g = (tap9 = b / NB8 -> dz = tap9 / 3, dx = tap9 % 3;
// channels 8 (b % NB8) .. + 7), column n = (dy = n >> 2, co = n & 3).  One scale for the tensor (exponent -> wexp[0]).
__global__ void __launch_bounds__(256) fm_pack_fwd_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int* __restrict__ wexp, int Cin, int Cout, int NSTEP) {
    __shared__ float red[4];
    const int ew = da_scale_exp(fm_tensor_absmax(w, 27 * Cin * Cout, red));
    if (threadIdx.x == 0) wexp[0] = ew;
    const float sc = da_pow2(ew);
    const int NB8 = Cin / 8, NBLK = 9 * NB8;
    for (int u = threadIdx.x; u < NSTEP * 64; u += 256) {
        const int lane = u & 63, st = u >> 6, n = lane & 15, g = lane >> 4;
        const int b = 4 * st + g, tap9 = b / NB8, cib = b - tap9 * NB8, dz = tap9 / 3, dx = tap9 - 3 * dz, dy = n >> 2, co = n & 3;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = (b < NBLK && dy < 3 && co < Cout) ? w[((size_t)((dz * 3 + dy) * 3 + dx) * Cin + cib * 8 + e) * Cout + co] : 0.f;
        uint2 h0, l0, h1, l1;
        da_split2(make_float4(v[0], v[1], v[2], v[3]), sc, h0, l0);
        da_split2(make_float4(v[4], v[5], v[6], v[7]), sc, h1, l1);
        uint4* o = reinterpret_cast<uint4*>(wp) + (size_t)st * 128 + lane;
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y); o[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

template <int C1, int C2>
__global__ void __launch_bounds__(256, 2) fm_fwd_kernel(FwdP p) {
    constexpr int Cin = C1 + C2, NB8 = Cin / 8, NBLK = 9 * NB8, NSTEP = (NBLK + 3) / 4;
    constexpr int VB = Cin * 2;                              // bytes per voxel and plane
    constexpr int PLANE = HV * VB;
    constexpr int Q1 = C1 / 4, Q2 = C2 / 4;
    constexpr int NIT1 = (HV * Q1 + 255) / 256, NIT2 = C2 ? (HV * Q2 + 255) / 256 : 0, NITS = NIT1 + NIT2;
    static_assert((Q1 & (Q1 - 1)) == 0 && (Q2 & (Q2 - 1)) == 0, "channel quads per voxel: powers of two");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* smax = reinterpret_cast<float*>(lds + 2 * PLANE);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4;
    const int zl = wave >> 1, half = wave & 1;
    // weights: 2 KB per K-step, L1 / L2-resident, fetched one step ahead (the whole operand in registers -- 56 of them -- spills beside the
    // 72 registers of the parked next tile)
    const __amdgpu_buffer_rsrc_t rsw = fm_rsrc(p.wp, (unsigned)(NSTEP * 2048));
    auto wb = [&](int s, int pl) -> f16x8 {
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, (unsigned)lane * 16u, (unsigned)((s * 2 + pl) * 1024), 0));
    };
    const int ew = p.wexp[0];
    int aoff[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        int b = 4 * s + g; if (b >= NBLK) b = NBLK - 1;      // (empty K blocks carry zero weights)
        const int tap9 = b / NB8, cib = b - tap9 * NB8, dz = tap9 / 3, dx = tap9 - 3 * dz;
        aoff[s] = ((dz * HY) * HX + dx) * VB + cib * 16;
    }
    const int abase = (((zl * HY) + 4 * half) * HX + i) * VB;
    // staging: tensor k, iteration it covers halo voxel hv = (tid + 256 it) / Qk, channel quad (tid + 256 it) % Qk
    float4 pre[NITS];
    auto issue = [&](int tile) {
        int n, z0, y0, x0;
        fm_tile(tile, p.ntz, p.nty, p.ntx, n, z0, y0, x0);
        const long long vox = (long long)p.D * p.H * p.W;
        const __amdgpu_buffer_rsrc_t r1 = fm_rsrc(p.in1 + (long long)n * vox * C1, (unsigned)(vox * C1 * 4));
        const __amdgpu_buffer_rsrc_t r2 = fm_rsrc(C2 ? p.in2 + (long long)n * vox * C2 : p.in1, (unsigned)(C2 ? vox * C2 * 4 : 0));
#pragma unroll
        for (int it = 0; it < NITS; ++it) {
            const bool second = it >= NIT1;
            const int Ck = second ? C2 : C1, Qk = second ? Q2 : Q1;
            const int idx = (int)threadIdx.x + (second ? it - NIT1 : it) * 256;
            const int q = idx & ((Qk ? Qk : 1) - 1);
            const int hv = idx / (Qk ? Qk : 1);
            const int hx = hv % HX, t = hv / HX, hy = t % HY, hz = t / HY;
            const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool inb = hv < HV && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            pre[it] = fm_load4(second ? r2 : r1, inb ? ((unsigned)((z * p.H + y) * p.W + x) * (unsigned)Ck + (unsigned)q * 4u) * 4u : 0xFFFFFFFFu);
        }
    };
    int Ecur = 0;
    auto stage = [&]() {                                     // (its barrier retires the tile in LDS)
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < NITS; ++it) m = da_absmax4(m, pre[it]);
        const int ex = da_scale_exp(da_block_max4(m, smax, wave, lane));
        const float sx = da_pow2(ex);
        Ecur = ex + ew;
#pragma unroll
        for (int it = 0; it < NITS; ++it) {
            const bool second = it >= NIT1;
            const int Qk = second ? Q2 : Q1;
            const int idx = (int)threadIdx.x + (second ? it - NIT1 : it) * 256;
            const int hv = idx / (Qk ? Qk : 1), q = idx & ((Qk ? Qk : 1) - 1);
            if (hv < HV) {
                uint2 h, l; da_split2(pre[it], sx, h, l);
                unsigned char* o = lds + hv * VB + ((second ? C1 : 0) + q * 4) * 2;
                *reinterpret_cast<uint2*>(o) = h; *reinterpret_cast<uint2*>(o + PLANE) = l;
            }
        }
    };
    float bv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) bv[c] = (p.bias && c < p.Cout) ? p.bias[c] : 0.f;
    auto ld = [&](int off) -> f16x8 { return *reinterpret_cast<const f16x8*>(lds + off); };
    const DaXcdItems TL = da_xcd_items(p.ntiles);            // an XCD's workgroups walk one contiguous eighth of the tile list together: neighbouring tiles' halo lines meet in ONE L2
    const int first = (int)TL.i, stride = (int)TL.step, tend = (int)TL.end;
    if (first < tend) { issue(first); stage(); }
    __syncthreads();
#pragma unroll 1
    for (int tile = first; tile < tend; tile += stride) {
        const bool more = tile + stride < tend;
        const int Eused = Ecur;
        if (more) issue(tile + stride);                      // next tile's loads fly under this tile's MFMAs
        f32x4 acc[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f16x8 A[3][2], An[3][2], B[2], Bn[2];
        B[0] = wb(0, 0); B[1] = wb(0, 1);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) A[r][pl] = ld(pl * PLANE + abase + aoff[0] + r * (HX * VB));
#pragma unroll
        for (int hs = 0; hs < 2 * NSTEP; ++hs) {             // half-steps: rows 0..2 / 3..5 of K-step hs / 2
            const int s = hs >> 1, r0 = (hs & 1) * 3;
            if ((hs & 1) == 0 && s + 1 < NSTEP) { Bn[0] = wb(s + 1, 0); Bn[1] = wb(s + 1, 1); }
            if (hs + 1 < 2 * NSTEP) {
                const int s1 = (hs + 1) >> 1, r1 = ((hs + 1) & 1) * 3;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) An[r][pl] = ld(pl * PLANE + abase + aoff[s1] + (r1 + r) * (HX * VB));
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) acc[r0 + r] = fm_mma3(acc[r0 + r], B, A[r]);
            __builtin_amdgcn_sched_barrier(0);
            if (hs + 1 < 2 * NSTEP) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) A[r][pl] = An[r][pl];
            }
            if ((hs & 1) == 1 && s + 1 < NSTEP) { B[0] = Bn[0]; B[1] = Bn[1]; }
        }
        // epilogue: out[y][co] = P[y][(0, co)] + P[y + 1][(1, co)] + P[y + 2][(2, co)]; lane group g holds the columns (dy = g, co = reg)
        {
            int n, z0, y0, x0;
            fm_tile(tile, p.ntz, p.nty, p.ntx, n, z0, y0, x0);
            const float inv1 = da_pow2(-(Eused / 2)), inv2 = da_pow2(-(Eused - Eused / 2));
            const long long osample = (long long)p.D * p.H * p.W * p.Cout;
            const __amdgpu_buffer_rsrc_t ro = fm_rsrc(p.out + (long long)n * osample, (unsigned)(osample * 4));
            const int z = z0 + zl, x = x0 + i;
#pragma unroll
            for (int yl = 0; yl < 4; ++yl) {
                const f32x4 t = g == 0 ? acc[yl] : (g == 1 ? acc[yl + 1] : (g == 2 ? acc[yl + 2] : (f32x4){0.f, 0.f, 0.f, 0.f}));
                float o[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float v = t[c];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    o[c] = da_act(v * inv1 * inv2 + bv[c], p.slope);
                }
                const int y = y0 + 4 * half + yl;
                const bool ok = g == 0 && z < p.D && y < p.H && x < p.W;
                const unsigned off = (unsigned)((z * p.H + y) * p.W + x) * (unsigned)p.Cout * 4u;      // (voxel index < 2^31 by da_conv3_flowmm_supported; the byte offset needs all 32 bits: unsigned arithmetic)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[c]), ro, (ok && c < p.Cout) ? off + 4 * c : 0xFFFFFFFFu, 0, 0);
            }
        }
        if (more) {
            stage();
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// data gradient
// ------------------------------------------------------------------------------------------------------------------------------
struct DgP {
    const float* dy; const unsigned char* wp; const int* wexp; float* dx1; float* dx2;
    int N, D, H, W, Cout, Cin, ntz, nty, ntx, ntiles;
};
constexpr int DG_STEPS = 4;                                  // K = 27 taps x 4 (cout padded) = 108 of 128

// packed weights of the data gradient: [step][N-tile][plane][lane][8]; lane (j, g): K = taps 8 step + 2 g + (e >> 2), cout e & 3; N = cin 16 nt + j
__global__ void __launch_bounds__(256) fm_pack_dgrad_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int* __restrict__ wexp, int Cin, int Cout, int NTN) {
    __shared__ float red[4];
    const int ew = da_scale_exp(fm_tensor_absmax(w, 27 * Cin * Cout, red));
    if (threadIdx.x == 0) wexp[0] = ew;
    const float sc = da_pow2(ew);
    for (int u = threadIdx.x; u < DG_STEPS * NTN * 64; u += 256) {
        const int lane = u & 63, blk = u >> 6, nt = blk % NTN, st = blk / NTN, j = lane & 15, g = lane >> 4;
        const int ci = nt * 16 + j;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int tap = 8 * st + 2 * g + (e >> 2), co = e & 3;
            v[e] = (tap < 27 && co < Cout && ci < Cin) ? w[((size_t)tap * Cin + ci) * Cout + co] : 0.f;
        }
        uint2 h0, l0, h1, l1;
        da_split2(make_float4(v[0], v[1], v[2], v[3]), sc, h0, l0);
        da_split2(make_float4(v[4], v[5], v[6], v[7]), sc, h1, l1);
        uint4* o = reinterpret_cast<uint4*>(wp) + (size_t)blk * 128 + lane;
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y); o[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

template <int C1, int C2>
__global__ void __launch_bounds__(256, 2) fm_dgrad_kernel(DgP p) {
    constexpr int Cin = C1 + C2, NTN = (Cin + 15) / 16;
    constexpr int PLANE = HV * 8;                            // [voxel][4 couts] fp16
    constexpr int NIT = (HV + 255) / 256;                    // 3
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* smax = reinterpret_cast<float*>(lds + 2 * PLANE);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4;
    const int zl = wave >> 1, half = wave & 1;
    f16x8 B[DG_STEPS][NTN][2];
#pragma unroll
    for (int s = 0; s < DG_STEPS; ++s)
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) B[s][nt][pl] = *(reinterpret_cast<const f16x8*>(p.wp) + ((s * NTN + nt) * 2 + pl) * 64 + lane);
    const int ew = p.wexp[0];
    // dY halo offsets of this lane group's two taps per step: output voxel v reads dY[v + 1 - d], halo origin at -1 -> halo index v + 2 - d
    int aoff[DG_STEPS][2];
#pragma unroll
    for (int s = 0; s < DG_STEPS; ++s)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int tap = 8 * s + 2 * g + k; if (tap > 26) tap = 26;     // (taps past 26 carry zero weights)
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            aoff[s][k] = (((2 - dz) * HY + (2 - dy)) * HX + (2 - dx)) * 8;
        }
    const int abase = ((zl * HY + 4 * half) * HX + i) * 8;
    float4 pre[NIT];
    auto issue = [&](int tile) {
        int n, z0, y0, x0;
        fm_tile(tile, p.ntz, p.nty, p.ntx, n, z0, y0, x0);
        const long long ysample = (long long)p.D * p.H * p.W * p.Cout;
        const __amdgpu_buffer_rsrc_t ry = fm_rsrc(p.dy + (long long)n * ysample, (unsigned)(ysample * 4));
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int hv = (int)threadIdx.x + it * 256;
            const int hx = hv % HX, t = hv / HX, hy = t % HY, hz = t / HY;
            const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool inb = hv < HV && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            const unsigned off = (unsigned)((z * p.H + y) * p.W + x) * (unsigned)p.Cout * 4u;      // (voxel index < 2^31 by da_conv3_flowmm_supported; the byte offset needs all 32 bits: unsigned arithmetic)
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = fm_load1(ry, (inb && c < p.Cout) ? off + 4 * c : 0xFFFFFFFFu);
            pre[it] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    int Ecur = 0;
    auto stage = [&]() {                                     // (its barrier retires the tile in LDS)
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) m = da_absmax4(m, pre[it]);
        const int ey = da_scale_exp(da_block_max4(m, smax, wave, lane));
        const float sy = da_pow2(ey);
        Ecur = ey + ew;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int hv = (int)threadIdx.x + it * 256;
            if (hv < HV) {
                uint2 h, l; da_split2(pre[it], sy, h, l);
                *reinterpret_cast<uint2*>(lds + hv * 8) = h; *reinterpret_cast<uint2*>(lds + PLANE + hv * 8) = l;
            }
        }
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto ld8 = [&](int pl, int row, int s) -> f16x8 {        // two taps x four couts: two 8-byte reads
        const unsigned char* a = lds + pl * PLANE + abase + row * (HX * 8);
        const u32x2 lo = *reinterpret_cast<const u32x2*>(a + aoff[s][0]), hi = *reinterpret_cast<const u32x2*>(a + aoff[s][1]);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
        return __builtin_bit_cast(f16x8, v);
    };
    const DaXcdItems TL = da_xcd_items(p.ntiles);            // an XCD's workgroups walk one contiguous eighth of the tile list together: neighbouring tiles' halo lines meet in ONE L2
    const int first = (int)TL.i, stride = (int)TL.step, tend = (int)TL.end;
    if (first < tend) { issue(first); stage(); }
    __syncthreads();
#pragma unroll 1
    for (int tile = first; tile < tend; tile += stride) {
        const bool more = tile + stride < tend;
        const int Eused = Ecur;
        if (more) issue(tile + stride);
        f32x4 acc[4][NTN];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) acc[r][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < DG_STEPS; ++s) {
            f16x8 A[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) A[r][pl] = ld8(pl, r, s);
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r][nt] = fm_mma3(acc[r][nt], B[s][nt], A[r]);
        }
        {
            int n, z0, y0, x0;
            fm_tile(tile, p.ntz, p.nty, p.ntx, n, z0, y0, x0);
            const float inv1 = da_pow2(-(Eused / 2)), inv2 = da_pow2(-(Eused - Eused / 2));
            const long long vox = (long long)p.D * p.H * p.W;
            const __amdgpu_buffer_rsrc_t r1 = fm_rsrc(p.dx1 + (long long)n * vox * C1, (unsigned)(vox * C1 * 4));
            const __amdgpu_buffer_rsrc_t r2 = fm_rsrc(C2 ? p.dx2 + (long long)n * vox * C2 : p.dx1, (unsigned)(C2 ? vox * C2 * 4 : 0));
            const int z = z0 + zl, x = x0 + i;
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) {
                const int c0 = nt * 16 + 4 * g;              // this lane's four input channels
                const bool to1 = c0 < C1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = y0 + 4 * half + r;
                    const bool ok = c0 < Cin && z < p.D && y < p.H && x < p.W;
                    const int v = (z * p.H + y) * p.W + x;
                    const unsigned off = to1 ? ((unsigned)v * (unsigned)C1 + (unsigned)c0) * 4u : ((unsigned)v * (unsigned)C2 + (unsigned)(c0 - C1)) * 4u;
                    fm_store4(to1 ? r1 : r2, ok ? off : 0xFFFFFFFFu, acc[r][nt] * inv1 * inv2);
                }
            }
        }
        if (more) {
            stage();
            __syncthreads();
        }
    }
}

template <typename K> int fm_set_lds(K kern, size_t bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
struct Plan { int ntz, nty, ntx, ntiles; };
Plan fm_plan(int N, int D, int H, int W) {
    Plan q;
    q.ntz = (D + TZ - 1) / TZ; q.nty = (H + TY - 1) / TY; q.ntx = (W + TX - 1) / TX;
    q.ntiles = N * q.ntz * q.nty * q.ntx;
    return q;
}

}  // namespace

// Shapes: split matrix mode, <= 3 output channels, the input channel splits instantiated below, samples below 4 GiB.
bool da_conv3_flowmm_supported(int C1, int C2, int Cout, int N, int D, int H, int W) {
    if (da_matrix_mode() != 2 || getenv("DA_NO_FLOWMM")) return false;
    if (Cout < 1 || Cout > 3) return false;
    if (!((C1 == 8 && C2 == 16) || (C1 == 16 && C2 == 16) || (C1 == 16 && C2 == 0) || (C1 == 8 && C2 == 8) || (C1 == 8 && C2 == 0))) return false;
    const long long vox = (long long)D * H * W;
    return vox * 16 * 4 < (1ll << 32) && (long long)N * vox < (1ll << 31);
}
size_t da_conv3_flowmm_ws_bytes() { return 65536; }          // exponent (256 B) + the packed weights (<= 36 KB)

#define FM_CASES(X) X(8, 16) X(16, 16) X(16, 0) X(8, 8) X(8, 0)

int da_conv3_flowmm_fwd(const float* in1, int C1, const float* in2, int C2, const float* w_tio, const float* bias, float* out,
                        int N, int D, int H, int W, int Cout, float slope, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!da_conv3_flowmm_supported(C1, C2, Cout, N, D, H, W)) return DA_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < da_conv3_flowmm_ws_bytes()) return DA_ERR_WS_SMALL;
    const Plan q = fm_plan(N, D, H, W);
    const int Cin = C1 + C2, NSTEP = (9 * (Cin / 8) + 3) / 4;
    FwdP p;
    // the packed operand: in the workspace, or in the caller's kept buffer (conv3d_internal.h: da_pp_lookup)
    const DaKeptPack kp = da_pp_lookup(w_tio, da_conv3_flowmm_ws_bytes(), DA_PP_FM_FWD);
    if (kp.only && !kp.buf) return 0;
    unsigned char* pk = kp.buf ? kp.buf : (unsigned char*)ws;
    p.in1 = in1; p.in2 = in2; p.wexp = (const int*)pk; p.wp = pk + 256; p.bias = bias; p.out = out;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cout = Cout; p.ntz = q.ntz; p.nty = q.nty; p.ntx = q.ntx; p.ntiles = q.ntiles; p.slope = slope;
    if (!kp.buf || kp.fill) {
        hipLaunchKernelGGL(fm_pack_fwd_kernel, dim3(1), dim3(256), 0, st, w_tio, (unsigned short*)(pk + 256), (int*)pk, Cin, Cout, NSTEP);
        DA_LAUNCH_CHECK();
    }
    if (kp.only) return 0;
    const int grid = q.ntiles < 512 ? q.ntiles : 512;
    const size_t shm = (size_t)2 * HV * Cin * 2 + 16;
#define X(a, b) if (C1 == a && C2 == b) { static bool set = false; if (!set) { const int e = fm_set_lds(fm_fwd_kernel<a, b>, shm); if (e) return e; set = true; } \
        hipLaunchKernelGGL((fm_fwd_kernel<a, b>), dim3(grid), dim3(256), shm, st, p); }
    FM_CASES(X)
#undef X
    DA_LAUNCH_CHECK();
    return 0;
}

int da_conv3_flowmm_dgrad(const float* dy, const float* w_tio, float* dx1, int C1, float* dx2, int C2,
                          int N, int D, int H, int W, int Cout, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!da_conv3_flowmm_supported(C1, C2, Cout, N, D, H, W)) return DA_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < da_conv3_flowmm_ws_bytes()) return DA_ERR_WS_SMALL;
    const Plan q = fm_plan(N, D, H, W);
    const int Cin = C1 + C2, NTN = (Cin + 15) / 16;
    DgP p;
    const DaKeptPack kp = da_pp_lookup(w_tio, da_conv3_flowmm_ws_bytes(), DA_PP_FM_DGRAD);
    if (kp.only && !kp.buf) return 0;
    unsigned char* pk = kp.buf ? kp.buf : (unsigned char*)ws;
    p.dy = dy; p.wexp = (const int*)pk; p.wp = pk + 256; p.dx1 = dx1; p.dx2 = dx2;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cout = Cout; p.Cin = Cin; p.ntz = q.ntz; p.nty = q.nty; p.ntx = q.ntx; p.ntiles = q.ntiles;
    if (!kp.buf || kp.fill) {
        hipLaunchKernelGGL(fm_pack_dgrad_kernel, dim3(1), dim3(256), 0, st, w_tio, (unsigned short*)(pk + 256), (int*)pk, Cin, Cout, NTN);
        DA_LAUNCH_CHECK();
    }
    if (kp.only) return 0;
    const int grid = q.ntiles < 512 ? q.ntiles : 512;
    const size_t shm = (size_t)2 * HV * 8 + 16;
#define X(a, b) if (C1 == a && C2 == b) hipLaunchKernelGGL((fm_dgrad_kernel<a, b>), dim3(grid), dim3(256), shm, st, p);
    FM_CASES(X)
#undef X
    DA_LAUNCH_CHECK();
    return 0;
}
