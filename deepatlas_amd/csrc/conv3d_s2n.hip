// Native stride-2 3x3x3 convolution (padding 1) for the registration encoder (voxel_morph.py:43-47, modules.py:48) in SPLIT matrix
// mode: forward, data gradient and weight gradient as implicit GEMMs on v_mfma_f32_16x16x32_f16 with both operands scaled by a power of
// two and split into two fp16 planes (x s = h + l, three partial products per multiply, fp32 accumulate: split_f16.h and conv3d_mfma.hip,
// "SP"; the scale is per staged tile for activations / gradients, per channel chunk (forward) or per tensor (data gradient) for weights).
//
// Why its own kernels: the space-to-depth route (conv3d_s2.hip) stages a 69 KB stride-1 halo tile per (parity, 16 channels) for 1 - 8 of
// the 27 taps; here one halo tile serves all 27 taps.
//   out[z][y][x][co] = b[co] + sum_{dz,dy,dx,ci} in[2z + dz - 1][2y + dy - 1][2x + dx - 1][ci] * W[dz][dy][dx][ci][co]
//
// LDS layout of an input halo tile (forward, weight gradient): [plane h|l][hz][hy][xpos][8 channels] fp16, where the halo's x axis is
// DE-INTERLEAVED by parity: halo column hx (input x = 2 x0 - 1 + hx) sits at xpos = hx / 2 for even hx, 17 + hx / 2 for odd hx.  The 16
// output voxels of an M-tile read columns hx = 2 i + dx, i.e. xpos = i (dx 0), 17 + i (dx 1), i + 1 (dx 2): consecutive lanes read
// consecutive 16-byte rows, exactly like the stride-1 kernels (no 2-way bank conflict from the 32-byte lane stride).
//
//  forward   M = output voxels (tile 2 x 4 x 16), N = Cout (wave = z plane x N-tile), K = 27 taps x 8 cin per channel chunk (7 K-steps
//            of 4 taps); the next chunk's global loads fly under the current chunk's MFMAs.  Two workgroups per CU.
//  dgrad     per input-parity class p = (pz, py, px) the gradient is a stride-1 conv of dY with (1 + pz)(1 + py)(1 + px) taps: one dY
//            halo tile (3 x 5 x 17 voxels, all Cout channels, three planes) feeds all 8 classes = all 27 taps; the four waves take the
//            class groups {111} {110,100,000} {101,010} {011,001} (8 / 7 / 6 / 6 taps, rotated per workgroup) with all 8 M-tiles each,
//            so a weight fragment is reused by 48 MFMAs.  Every input voxel is written exactly once (no zero fill, no atomics).
//  wgrad     M = 2 taps x 8 cin, N = Cout, K = voxels (16 along x times the tile's two z planes per MFMA, ds_read_b64_tr_b16 transposes
//            both operands out of their channel-contiguous tiles); every wave owns one output row of the 2 x 4 x 16 tile and all 14 tap
//            pairs x 2 N-tiles (112 accumulator registers) over a persistent walk; per-workgroup partial dW, reduced in double, in a
//            fixed order, by da_reduce_partials.
// Roofline: MFMA-bound in FLOPs (2 * 27 * Cin * Cout per OUTPUT voxel) only above ~300 TFLOP/s; at the registration net's sizes
// (16 -> 32 at 160 x 192 x 160: 17 GFLOP, 315 MB in + 79 MB out) the forward and the data gradient are HBM/L2-bound.
#include "common.h"
#include "conv3d_internal.h"
#include "split_f16.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TZ = 2, TY = 4, TX = 16;                          // output tile (forward / wgrad), dY-grid tile (dgrad)
constexpr int HZ = 2 * TZ + 1, HY = 2 * TY + 1, HXV = 2 * TX + 1;     // input halo voxels: 5 x 9 x 33
constexpr int XS = 33;                                          // xpos slots per halo row (17 even + 16 odd)
constexpr int PLANE_V = HZ * HY * XS;                           // voxel slots per plane
constexpr int PLANE_B = PLANE_V * 16;                           // bytes per plane (8 bf16 per voxel)
constexpr int NSTEPS = 7;                                       // forward K-steps per 8-channel chunk: 4 taps each (28 slots, the last one empty)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t s2n_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 s2n_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void s2n_store4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r, off, 0, 0);
}
__device__ __forceinline__ float s2n_qx1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }   // lane ^ 1
__device__ __forceinline__ float s2n_qx2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }   // lane ^ 2
// 4 x 4 transpose across a lane quad: in = (4 voxels x 1 channel) per lane (MFMA C layout: row = 4 g + reg), out = (1 voxel x 4 channels)
__device__ __forceinline__ f32x4 s2n_quad_transpose(f32x4 a, int q) {
    float t0 = a[0], t1 = a[1], t2 = a[2], t3 = a[3];
    {
        const bool odd = (q & 1) != 0;
        const float s01 = odd ? t0 : t1, s23 = odd ? t2 : t3;
        const float r01 = s2n_qx1(s01), r23 = s2n_qx1(s23);
        if (odd) { t0 = r01; t2 = r23; } else { t1 = r01; t3 = r23; }
    }
    {
        const bool hi2 = (q & 2) != 0;
        const float s02 = hi2 ? t0 : t2, s13 = hi2 ? t1 : t3;
        const float r02 = s2n_qx2(s02), r13 = s2n_qx2(s13);
        if (hi2) { t0 = r02; t1 = r13; } else { t2 = r02; t3 = r13; }
    }
    return (f32x4){t0, t1, t2, t3};
}
__device__ __forceinline__ int s2n_xcd_remap(int bid, int nwg) {        // consecutive tiles on the same XCD (shared halos hit that XCD's L2)
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, loc = bid / 8;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}
// products of a split multiply, small terms first: (a plane, b plane) = (h,l) (l,h) (h,h)
#define S2N_PLANE_PAIRS constexpr int kPA[3] = {0, 1, 0}, kPB[3] = {1, 0, 0}
constexpr int NPLN = 2;                                         // operand planes (h, l)
// largest |w| of a whole weight tensor (n floats, n % 4 == 0, 16-byte aligned), by one 256-thread workgroup; `red` = 4 floats of LDS
__device__ __forceinline__ float s2n_tensor_absmax(const float* __restrict__ w, int n, float* red) {
    float m = 0.f;
#pragma unroll 4
    for (int q = threadIdx.x; q < n / 4; q += 256) m = da_absmax4(m, reinterpret_cast<const float4*>(w)[q]);
    return da_block_max4(m, red, (int)threadIdx.x >> 6, (int)threadIdx.x & 63);
}

// ------------------------------------------------------------------------------------------------------------------------------
// halo staging shared by forward and weight gradient: 5 x 9 x 33 voxels x 8 channels (two 16-byte quads per voxel), 12 iterations
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int HALO_Q = HZ * HY * HXV * 2;                       // float4 quads per chunk tile
constexpr int HALO_NIT = (HALO_Q + 255) / 256;                  // 12

struct HaloMap {
    unsigned pk[HALO_NIT];        // (hz << 16 | hy << 8 | hx) of this thread's voxel per iteration; 0xFFFFxxxx = past the tile
    int ldsq[HALO_NIT];           // uint2 index inside a plane of the staged quad
    int c4;
    __device__ __forceinline__ void init() {
        c4 = (int)threadIdx.x & 1;
#pragma unroll
        for (int it = 0; it < HALO_NIT; ++it) {
            const int hv = ((int)threadIdx.x + it * 256) >> 1;
            const int hx = hv % HXV, t = hv / HXV, hy = t % HY, hz = t / HY;
            const bool ok = hv < HZ * HY * HXV;
            pk[it] = ok ? ((unsigned)hz << 16 | (unsigned)hy << 8 | (unsigned)hx) : 0xFFFF0000u;
            const int xpos = (hx & 1) ? 17 + (hx >> 1) : (hx >> 1);
            ldsq[it] = ok ? (((hz * HY + hy) * XS + xpos) * 2 + c4) : -1;
        }
    }
    // byte offset of iteration `it` inside one sample of an [D][H][W][Cs] tensor, halo origin (zb, yb, xb) = 2 * tile origin - 1
    __device__ __forceinline__ unsigned offset(int it, int zb, int yb, int xb, int D, int H, int W, int Cs, int choff) const {
        const int hz = (int)(pk[it] >> 16), hy = (int)((pk[it] >> 8) & 255u), hx = (int)(pk[it] & 255u);
        const int z = zb + hz, y = yb + hy, x = xb + hx;
        const bool inb = (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && hz != 0xFFFF;
        return inb ? (unsigned)((((z * H + y) * W + x) * Cs + choff + c4 * 4) * 4) : 0xFFFFFFFFu;
    }
    __device__ __forceinline__ void write(unsigned char* lds, const float4* pre, const float scale) const {
#pragma unroll
        for (int it = 0; it < HALO_NIT; ++it) {
            if (ldsq[it] >= 0) {
                uint2 h, l; da_split2(pre[it], scale, h, l);
                uint2* p = reinterpret_cast<uint2*>(lds) + ldsq[it];
                p[0] = h; p[PLANE_B / 8] = l;
            }
        }
    }
    __device__ __forceinline__ float absmax(const float4* pre) const {      // (quads past the tile were loaded out of range: zeros)
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < HALO_NIT; ++it) m = da_absmax4(m, pre[it]);
        return m;
    }
};

// ------------------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------------------
struct FwdP {
    const float* in; const unsigned char* wp; const int* wexp; const float* bias; float* out;
    int N, D, H, W, Cin, Cout, Do, Ho, Wo, ntz, nty, ntx, nchunks, NT;
    float slope;
};

// packed B operand of the forward: [chunk][step][N-tile][plane][lane][8 fp16]; lane (g, j): tap 4 step + g, cin chunk * 8 + e, cout 16 nt + j.
// Grid (chunks, PY): every workgroup of a chunk finds the chunk's largest |w| (its scale exponent goes to wexp[chunk]) and packs its share.
__global__ void __launch_bounds__(256) s2n_pack_fwd_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int* __restrict__ wexp, int Cin, int Cout, int NT) {
    __shared__ float red[4];
    const int ch = blockIdx.x;
    float m = 0.f;
    const int qc = Cout / 4, nq = 27 * 8 * qc;                  // runs of Cout consecutive couts per (tap, ci); Cout % 32 == 0
#pragma unroll 4
    for (int q = threadIdx.x; q < nq; q += 256) {
        const int r = q / qc, c4 = q - r * qc;
        m = da_absmax4(m, *reinterpret_cast<const float4*>(w + ((size_t)(r >> 3) * Cin + ch * 8 + (r & 7)) * Cout + c4 * 4));
    }
    const int ew = da_scale_exp(da_block_max4(m, red, (int)threadIdx.x >> 6, (int)threadIdx.x & 63));
    if (blockIdx.y == 0 && threadIdx.x == 0) wexp[ch] = ew;
    const float sc = da_pow2(ew);
    const int units = NSTEPS * NT * 64;
    for (int u = blockIdx.y * 256 + threadIdx.x; u < units; u += gridDim.y * 256) {
        const int lane = u & 63, nt = (u >> 6) % NT, st = (u >> 6) / NT;
        const int tap = 4 * st + (lane >> 4), co = nt * 16 + (lane & 15);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (tap < 27 && co < Cout) ? w[((size_t)tap * Cin + ch * 8 + e) * Cout + co] : 0.f;
        uint2 h0, l0, h1, l1;
        da_split2(make_float4(v[0], v[1], v[2], v[3]), sc, h0, l0);
        da_split2(make_float4(v[4], v[5], v[6], v[7]), sc, h1, l1);
        uint4* o = reinterpret_cast<uint4*>(wp + ((size_t)((ch * NSTEPS + st) * NT + nt) * NPLN) * 512) + lane;
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y); o[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

__global__ void __launch_bounds__(256, 2) s2n_fwd_kernel(FwdP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4, q = lane & 3, a4 = (lane & 15) >> 2;
    const int zi = wave >> 1;                                   // output z plane of the tile
    const int nt = blockIdx.y * 2 + (wave & 1);                 // this wave's N-tile
    // tile of this workgroup: z fastest, then y, then x (z / y neighbours share the most halo and run back to back on one XCD)
    int t = s2n_xcd_remap(blockIdx.x, gridDim.x);
    const int tz = t % p.ntz; t /= p.ntz;
    const int ty = t % p.nty; t /= p.nty;
    const int tx = t % p.ntx; const int n = t / p.ntx;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
    const long long sample = (long long)p.D * p.H * p.W * p.Cin;
    const __amdgpu_buffer_rsrc_t rs = s2n_rsrc(p.in + (long long)n * sample, (unsigned)(sample * sizeof(float)));
    const __amdgpu_buffer_rsrc_t rsw = s2n_rsrc(p.wp, (unsigned)((size_t)p.nchunks * NSTEPS * p.NT * NPLN * 1024));
    HaloMap hm; hm.init();
    float4 pre[HALO_NIT];
    auto issue = [&](int ch) {
#pragma unroll
        for (int it = 0; it < HALO_NIT; ++it) pre[it] = s2n_load4(rs, hm.offset(it, 2 * z0 - 1, 2 * y0 - 1, 2 * x0 - 1, p.D, p.H, p.W, p.Cin, ch * 8));
    };
    // per-lane tap offsets (bytes) of the K-steps: lane group g -> tap 4 s + g
    int aoff[NSTEPS];
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
        int tap = 4 * s + g; if (tap > 26) tap = 26;            // (slot 27 carries zero weights)
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        const int xp = dx == 0 ? 0 : (dx == 1 ? 17 : 1);
        aoff[s] = ((dz * HY + dy) * XS + xp) * 16;
    }
    const int abase = ((2 * zi * HY) * XS + i) * 16;
    auto wb = [&](int ch, int s, int pl) -> f16x8 {
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, (unsigned)lane * 16u, (unsigned)((((ch * NSTEPS + s) * p.NT + nt) * NPLN + pl) * 1024), 0));
    };
    auto ld = [&](int off) -> f16x8 { return *reinterpret_cast<const f16x8*>(lds + off); };
    // scale bookkeeping (wave-uniform; conv3d_mfma.hip "SP"): the accumulators hold (true sums) x 2^Eacc, chunk ch was staged at 2^(E - wexp[ch]),
    // E of a later chunk is capped at 40 above the smallest E so far
    float* smax = reinterpret_cast<float*>(lds + NPLN * PLANE_B);
    int Eacc = 0, Emin = 0;
    auto stage = [&](int ch) -> int {                            // publish the tile's largest magnitude, barrier, split + write; returns the chunk's E
        const float m = da_block_max4(hm.absmax(pre), smax, wave, lane);
        const int ew = p.wexp[ch];
        int E = da_scale_exp(m) + ew;
        if (ch != 0) E = min(E, Emin + 40);
        Emin = (ch == 0) ? E : min(Emin, E);
        hm.write(lds, pre, da_pow2(E - ew));
        return E;
    };

    S2N_PLANE_PAIRS;
    f32x4 acc[TY];
#pragma unroll
    for (int r = 0; r < TY; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    issue(0);
    int Ecur = stage(0);
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < p.nchunks; ++ch) {
        const bool more = ch + 1 < p.nchunks;
        if (more) issue(ch + 1);                                 // next chunk's loads fly under this chunk's MFMAs
        {
            const float f = da_acc_factor(Ecur - Eacc);                // the running sums into this chunk's unit (exact)
#pragma unroll
            for (int r = 0; r < TY; ++r) acc[r] = acc[r] * f;
            Eacc = Ecur;
        }
        f16x8 B[NPLN], Bn[NPLN], A[TY][NPLN], An[TY][NPLN];
#pragma unroll
        for (int pl = 0; pl < NPLN; ++pl) B[pl] = wb(ch, 0, pl);
#pragma unroll
        for (int r = 0; r < TY; ++r)
#pragma unroll
            for (int pl = 0; pl < NPLN; ++pl) A[r][pl] = ld(pl * PLANE_B + abase + aoff[0] + r * (2 * XS * 16));
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            if (s + 1 < NSTEPS) {
#pragma unroll
                for (int pl = 0; pl < NPLN; ++pl) Bn[pl] = wb(ch, s + 1, pl);
#pragma unroll
                for (int r = 0; r < TY; ++r)
#pragma unroll
                    for (int pl = 0; pl < NPLN; ++pl) An[r][pl] = ld(pl * PLANE_B + abase + aoff[s + 1] + r * (2 * XS * 16));
            }
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int r = 0; r < TY; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[r][kPA[pr]], B[kPB[pr]], acc[r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < NSTEPS) {
#pragma unroll
                for (int pl = 0; pl < NPLN; ++pl) B[pl] = Bn[pl];
#pragma unroll
                for (int r = 0; r < TY; ++r)
#pragma unroll
                    for (int pl = 0; pl < NPLN; ++pl) A[r][pl] = An[r][pl];
            }
        }
        if (more) {
            Ecur = stage(ch + 1);                                // (its barrier: every wave is done reading this chunk's tile)
            __syncthreads();
        }
    }
    {
        const float inv1 = da_pow2(-(Eacc / 2)), inv2 = da_pow2(-(Eacc - Eacc / 2));      // back to the true unit
#pragma unroll
        for (int r = 0; r < TY; ++r) acc[r] = acc[r] * inv1 * inv2;
    }
    // epilogue: lane -> voxel x0 + 4 g + q, couts 16 nt + 4 a4 .. + 3
    const int co0 = nt * 16 + 4 * a4;
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = (p.bias && co0 + j < p.Cout) ? p.bias[co0 + j] : 0.f;
    const long long osample = (long long)p.Do * p.Ho * p.Wo * p.Cout;
    const __amdgpu_buffer_rsrc_t ro = s2n_rsrc(p.out + (long long)n * osample, (unsigned)(osample * sizeof(float)));
    const int z = z0 + zi, x = x0 + 4 * g + q;
    const bool ok0 = z < p.Do && x < p.Wo && co0 + 3 < p.Cout;
#pragma unroll
    for (int r = 0; r < TY; ++r) {
        const f32x4 v = s2n_quad_transpose(acc[r], q);
        const f32x4 o = {da_act(v[0] + bv[0], p.slope), da_act(v[1] + bv[1], p.slope), da_act(v[2] + bv[2], p.slope), da_act(v[3] + bv[3], p.slope)};
        const int y = y0 + r;
        s2n_store4(ro, (ok0 && y < p.Ho) ? (unsigned)((((z * p.Ho + y) * p.Wo + x) * p.Cout + co0) * 4) : 0xFFFFFFFFu, o);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// data gradient
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int GZ = TZ + 1, GY = TY + 1, GX = TX + 1;           // dY halo: 3 x 5 x 17 voxels (one extra on the high side)
constexpr int GV = GZ * GY * GX;                                // 255

struct DgP {
    const float* dy; const unsigned char* wp; const int* wexp; float* dx;
    int N, D, H, W, Cin, Cout, Do, Ho, Wo, ntz, nty, ntx, KS, NTN;   // KS = Cout / 32 K-steps per tap, NTN = Cin / 16 N-tiles
};

__host__ __device__ inline int s2n_class_taps(int c) { return (1 + ((c >> 2) & 1)) * (1 + ((c >> 1) & 1)) * (1 + (c & 1)); }
// tap k (0 .. ntaps-1) of parity class c: original tap index (dz*9 + dy*3 + dx) and the source offset (oz, oy, ox) in the dY grid
__host__ __device__ inline void s2n_class_tap(int c, int k, int& tap, int& oz, int& oy, int& ox) {
    const int pz = (c >> 2) & 1, py = (c >> 1) & 1, px = c & 1;
    const int kx = px ? (k & 1) : 0; const int k1 = px ? (k >> 1) : k;
    const int ky = py ? (k1 & 1) : 0; const int kz = py ? (k1 >> 1) : k1;
    const int dz = pz ? (kz == 0 ? 0 : 2) : 1, dyy = py ? (ky == 0 ? 0 : 2) : 1, dx = px ? (kx == 0 ? 0 : 2) : 1;
    oz = (pz && kz == 0) ? 1 : 0; oy = (py && ky == 0) ? 1 : 0; ox = (px && kx == 0) ? 1 : 0;
    tap = dz * 9 + dyy * 3 + dx;
}
__host__ __device__ inline int s2n_class_base(int c) { int b = 0; for (int k = 0; k < c; ++k) b += s2n_class_taps(k); return b; }    // in taps

// packed B operand of the data gradient: [class-major step q = (class base + k) * KS + ks][N-tile][plane][lane][8 fp16];
// lane (g, j): K index = cout ks * 32 + g * 8 + e, N index = cin 16 nt + j.  ONE scale for the whole tensor (every tap and channel of a
// parity class accumulates into the same sums): each workgroup finds it (the tensor is a few hundred KB, L2-resident), block 0 stores its
// exponent in wexp[0].
__global__ void __launch_bounds__(256) s2n_pack_dgrad_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int* __restrict__ wexp, int Cin, int Cout, int KS, int NTN) {
    __shared__ float red[4];
    const int ew = da_scale_exp(s2n_tensor_absmax(w, 27 * Cin * Cout, red));
    if (blockIdx.x == 0 && threadIdx.x == 0) wexp[0] = ew;
    const float sc = da_pow2(ew);
    const int units = 27 * KS * NTN * 64;
    for (int u = blockIdx.x * 256 + threadIdx.x; u < units; u += gridDim.x * 256) {
        const int lane = u & 63, blk = u >> 6;
        const int nt = blk % NTN, ks = (blk / NTN) % KS, tq = blk / (NTN * KS);      // tq = class base + k, 0 .. 26
        int c = 0, base = 0;
        while (base + s2n_class_taps(c) <= tq) { base += s2n_class_taps(c); ++c; }
        int tap, oz, oy, ox; s2n_class_tap(c, tq - base, tap, oz, oy, ox);
        const int g = lane >> 4, j = lane & 15;
        const int co0 = ks * 32 + g * 8, ci = nt * 16 + j;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (ci < Cin && co0 + 7 < Cout) {                         // 8 consecutive couts
            const float* q = w + ((size_t)tap * Cin + ci) * Cout + co0;
            v0 = *reinterpret_cast<const float4*>(q); v1 = *reinterpret_cast<const float4*>(q + 4);
        }
        uint2 h0, l0, h1, l1;
        da_split2(v0, sc, h0, l0); da_split2(v1, sc, h1, l1);
        uint4* o = reinterpret_cast<uint4*>(wp + (size_t)blk * NPLN * 512) + lane;
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y); o[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

template <int NTN>
__global__ void __launch_bounds__(256, 2) s2n_dgrad_kernel(DgP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4, q = lane & 3, a4 = (lane & 15) >> 2;
    int t = s2n_xcd_remap(blockIdx.x, gridDim.x);
    const int tz = t % p.ntz; t /= p.ntz;
    const int ty = t % p.nty; t /= p.nty;
    const int tx = t % p.ntx; const int n = t / p.ntx;
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
    const int nch = p.Cout / 8;                                  // 8-channel chunks of dY: LDS [plane][chunk][voxel][8]
    const int planeB = nch * GV * 16;
    // ---- stage the dY halo tile (all channels) at the tile's own scale, split into two planes.  All of it is parked in registers first
    // (<= 16 quads per thread for Cout <= 64: the accumulators are not live yet) because the scale needs the tile's largest magnitude.
    int E = 0;
    {
        const long long ysample = (long long)p.Do * p.Ho * p.Wo * p.Cout;
        const __amdgpu_buffer_rsrc_t ry = s2n_rsrc(p.dy + (long long)n * ysample, (unsigned)(ysample * sizeof(float)));
        const int qpv = p.Cout / 4;                              // quads per voxel
        const int total = GV * qpv;
        constexpr int MAXU = 16;                                 // GV * 16 quads / 256 threads
        float4 v[MAXU]; int li[MAXU];
        float m = 0.f;
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
            const int idx = u * 256 + (int)threadIdx.x;
            const int c4 = idx % qpv, hv = idx / qpv;
            const int hx = hv % GX, t2 = hv / GX, hy = t2 % GY, hz = t2 / GY;
            const int z = z0 + hz, y = y0 + hy, x = x0 + hx;
            const bool inb = idx < total && z < p.Do && y < p.Ho && x < p.Wo;
            v[u] = s2n_load4(ry, inb ? (unsigned)((((z * p.Ho + y) * p.Wo + x) * p.Cout + c4 * 4) * 4) : 0xFFFFFFFFu);
            li[u] = idx < total ? (((c4 >> 1) * GV + hv) * 2 + (c4 & 1)) : -1;
        }
#pragma unroll
        for (int u = 0; u < MAXU; ++u) m = da_absmax4(m, v[u]);
        float* smax = reinterpret_cast<float*>(lds + NPLN * planeB);
        const int ey = da_scale_exp(da_block_max4(m, smax, wave, lane));
        E = ey + p.wexp[0];
        const float sy = da_pow2(ey);
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
            if (li[u] >= 0) {
                uint2 h, l; da_split2(v[u], sy, h, l);
                uint2* o = reinterpret_cast<uint2*>(lds) + li[u];
                o[0] = h; o[planeB / 8] = l;
            }
        }
    }
    __syncthreads();
    const float inv1 = da_pow2(-(E / 2)), inv2 = da_pow2(-(E - E / 2));      // the sums back to the true unit (two exact factors)
    const __amdgpu_buffer_rsrc_t rsw = s2n_rsrc(p.wp, (unsigned)((size_t)27 * p.KS * NTN * NPLN * 1024));
    const long long xsample = (long long)p.D * p.H * p.W * p.Cin;
    const __amdgpu_buffer_rsrc_t rx = s2n_rsrc(p.dx + (long long)n * xsample, (unsigned)(xsample * sizeof(float)));
    // class groups per wave role (rotated by workgroup so that the 8-tap class does not always sit on the same SIMD)
    const int role = (wave + blockIdx.x) & 3;
    const int ncls = role == 0 ? 1 : (role == 1 ? 3 : 2);
    auto cls_of = [&](int k) -> int { return role == 0 ? 7 : (role == 1 ? (k == 0 ? 6 : (k == 1 ? 4 : 0)) : (role == 2 ? (k == 0 ? 5 : 2) : (k == 0 ? 3 : 1))); };
    auto ld = [&](int off) -> f16x8 { return *reinterpret_cast<const f16x8*>(lds + off); };
    S2N_PLANE_PAIRS;
#pragma unroll 1
    for (int kc = 0; kc < ncls; ++kc) {
        const int c = cls_of(kc);
        const int pz = (c >> 2) & 1, py = (c >> 1) & 1, px = c & 1;
        const int ntaps = s2n_class_taps(c), cbase = s2n_class_base(c);
        f32x4 acc[TZ * TY][NTN];
#pragma unroll
        for (int m = 0; m < TZ * TY; ++m)
#pragma unroll
            for (int nn = 0; nn < NTN; ++nn) acc[m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int k = 0; k < ntaps; ++k) {
            int tap, oz, oy, ox; s2n_class_tap(c, k, tap, oz, oy, ox);
#pragma unroll 1
            for (int ks = 0; ks < p.KS; ++ks) {
                const int step = (cbase + k) * p.KS + ks;
                f16x8 B[NTN][NPLN];
#pragma unroll
                for (int nn = 0; nn < NTN; ++nn)
#pragma unroll
                    for (int pl = 0; pl < NPLN; ++pl)
                        B[nn][pl] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, (unsigned)lane * 16u, (unsigned)(((step * NTN + nn) * NPLN + pl) * 1024), 0));
                const int abase = (((ks * 4 + g) * GV) + (oz * GY + oy) * GX + ox + i) * 16;
#pragma unroll
                for (int mz = 0; mz < TZ; ++mz) {
                    f16x8 A[TY][NPLN];
#pragma unroll
                    for (int my = 0; my < TY; ++my)
#pragma unroll
                        for (int pl = 0; pl < NPLN; ++pl) A[my][pl] = ld(pl * planeB + abase + ((mz * GY + my) * GX) * 16);
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int nn = 0; nn < NTN; ++nn)
#pragma unroll
                            for (int my = 0; my < TY; ++my)
                                acc[mz * TY + my][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[my][kPA[pr]], B[nn][kPB[pr]], acc[mz * TY + my][nn], 0, 0, 0);
                }
            }
        }
        // stores: input voxel (2 z2 + pz, 2 y2 + py, 2 x2 + px), x2 = x0 + 4 g + q, channels 16 nn + 4 a4 .. + 3
        const int xin = 2 * (x0 + 4 * g + q) + px;
#pragma unroll
        for (int mz = 0; mz < TZ; ++mz) {
            const int zin = 2 * (z0 + mz) + pz;
#pragma unroll
            for (int my = 0; my < TY; ++my) {
                const int yin = 2 * (y0 + my) + py;
                const bool ok = zin < p.D && yin < p.H && xin < p.W;
#pragma unroll
                for (int nn = 0; nn < NTN; ++nn) {
                    const f32x4 v = s2n_quad_transpose(acc[mz * TY + my][nn] * inv1 * inv2, q);
                    s2n_store4(rx, ok ? (unsigned)((((zin * p.H + yin) * p.W + xin) * p.Cin + nn * 16 + 4 * a4) * 4) : 0xFFFFFFFFu, v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------------------------------------
struct WgP {
    const float* in; const float* dy; float* partial;
    int N, D, H, W, Cin, Cout, Do, Ho, Wo, ntz, nty, ntx, ntiles, nslabs, O;
};
constexpr int WG_SLOTS = 14;                                    // tap pairs (2 k, 2 k + 1); the upper half of slot 13 is empty
constexpr int YV = TZ * TY * TX;                                // 128 dY voxels per tile
constexpr int YPLANE_B = YV * 32 * 2;                           // bytes per dY plane: [voxel][32 cout] bf16

__global__ void __launch_bounds__(256, 1) s2n_wgrad_kernel(WgP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* ldsY = lds + NPLN * PLANE_B;
    typedef s16x4 __attribute__((address_space(3))) * lds_frag_ptr;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4, q = i & 3, vq = i >> 2;
    const int ch = blockIdx.y, cg = blockIdx.z;                 // 8-channel input chunk, 32-channel output group
    const int zi = g >> 1;                                       // z plane of the tile this lane group's 8 voxels belong to
    // transposing reads (ds_read_b64_tr_b16): inside a 16-lane group lane (vq, q) supplies the address of (voxel vq of 4, 4-channel quad q)
    // and receives 4 voxels of channel 4 q' + e ... i.e. fragment row i = (tap half i >> 3, cin i & 7) for x, cout i for dY.
    auto tr8 = [&](const unsigned char* a, int step_bytes) -> f16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)a);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)(a + step_bytes));
        return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // x fragment of tap-pair slot s for this wave's output row: lane supplies voxel (z plane zi, row, x2 = 8 (g & 1) + vq [+ 4]) of tap
    // 2 s + (q >> 1), channel quad q & 1
    int aoff[WG_SLOTS];
#pragma unroll
    for (int s = 0; s < WG_SLOTS; ++s) {
        int tap = 2 * s + (q >> 1); if (tap > 26) tap = 26;     // (slot 13, upper half: re-reads tap 26; its rows are never written)
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        const int xp = dx == 0 ? 0 : (dx == 1 ? 17 : 1);
        aoff[s] = ((((2 * zi + dz) * HY + 2 * wave + dy) * XS + xp + 8 * (g & 1) + vq) * 8 + (q & 1) * 4) * 2;
    }
    const int yoff = (((zi * TY + wave) * TX + 8 * (g & 1) + vq) * 32 + q * 4) * 2;
    struct F3 { f16x8 p[NPLN]; };
    auto loadF = [&](int s) -> F3 {
        F3 f;
#pragma unroll
        for (int pl = 0; pl < NPLN; ++pl) f.p[pl] = tr8(lds + pl * PLANE_B + aoff[s], 4 * 8 * 2);
        return f;
    };
    auto loadY = [&](int nn) -> F3 {
        F3 f;
#pragma unroll
        for (int pl = 0; pl < NPLN; ++pl) f.p[pl] = tr8(ldsY + pl * YPLANE_B + yoff + nn * 32, 4 * 32 * 2);
        return f;
    };
    f32x4 acc[WG_SLOTS][2];
#pragma unroll
    for (int s = 0; s < WG_SLOTS; ++s) { acc[s][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[s][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    HaloMap hm; hm.init();
    float4 preA[HALO_NIT], preY[4];
    const int yv0 = (int)threadIdx.x >> 3, yq = (int)threadIdx.x & 7;      // dY staging: iteration u covers voxel yv0 + 32 u, cout quad yq (of 8)
    auto issue = [&](int pos) {
        int t = pos;
        const int tz = t % p.ntz; t /= p.ntz;
        const int ty = t % p.nty; t /= p.nty;
        const int tx = t % p.ntx; const int n = t / p.ntx;
        const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * TX;
        const long long sample = (long long)p.D * p.H * p.W * p.Cin;
        const __amdgpu_buffer_rsrc_t rs = s2n_rsrc(p.in + (long long)n * sample, (unsigned)(sample * sizeof(float)));
#pragma unroll
        for (int it = 0; it < HALO_NIT; ++it) preA[it] = s2n_load4(rs, hm.offset(it, 2 * z0 - 1, 2 * y0 - 1, 2 * x0 - 1, p.D, p.H, p.W, p.Cin, ch * 8));
        const long long ysample = (long long)p.Do * p.Ho * p.Wo * p.Cout;
        const __amdgpu_buffer_rsrc_t ry = s2n_rsrc(p.dy + (long long)n * ysample, (unsigned)(ysample * sizeof(float)));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int v = yv0 + 32 * u;
            const int vx = v & 15, vy = (v >> 4) & 3, vz = v >> 6;
            const int z = z0 + vz, y = y0 + vy, x = x0 + vx, co = cg * 32 + yq * 4;
            const bool inb = z < p.Do && y < p.Ho && x < p.Wo && co < p.Cout;
            preY[u] = s2n_load4(ry, inb ? (unsigned)((((z * p.Ho + y) * p.Wo + x) * p.Cout + co) * 4) : 0xFFFFFFFFu);
        }
    };
    // scale bookkeeping (wave-uniform; conv3d_mfma.hip, conv3_split_wgrad_kernel): the accumulators hold (true sums) x 2^Eacc; a tile is staged at
    // E = x exponent + dY exponent, capped at 40 above the smallest E of this workgroup's walk so far
    float* smax = reinterpret_cast<float*>(ldsY + NPLN * YPLANE_B);      // [2][4]
    int Eacc = 0, Emin = 0, Enext = 0; bool first_tile = true;
    auto write_lds = [&]() {                                     // (starts with the barrier that retires the tile in LDS)
        float my = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) my = da_absmax4(my, preY[u]);
        const float ma = da_wave_max_nonneg(hm.absmax(preA));
        my = da_wave_max_nonneg(my);
        if (lane == 0) { smax[wave] = ma; smax[4 + wave] = my; }
        __syncthreads();
        const float4 a4 = *reinterpret_cast<const float4*>(smax), y4 = *reinterpret_cast<const float4*>(smax + 4);
        const int ea = da_scale_exp(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fmaxf(fmaxf(a4.x, a4.y), fmaxf(a4.z, a4.w))))));
        const int ey = da_scale_exp(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fmaxf(fmaxf(y4.x, y4.y), fmaxf(y4.z, y4.w))))));
        int E = ea + ey;
        if (!first_tile) E = min(E, Emin + 40);
        Emin = first_tile ? E : min(Emin, E);
        first_tile = false;
        Enext = E;
        const float sy = da_pow2(ey);
        hm.write(lds, preA, da_pow2(E - ey));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint2 h, l; da_split2(preY[u], sy, h, l);
            uint2* o = reinterpret_cast<uint2*>(ldsY) + (yv0 + 32 * u) * 8 + yq;
            o[0] = h; o[YPLANE_B / 8] = l;
        }
    };
    S2N_PLANE_PAIRS;
    const int slab = blockIdx.x;
    const int cnt = (p.ntiles > slab) ? (p.ntiles - slab + p.nslabs - 1) / p.nslabs : 0;
    if (cnt > 0) { issue(slab); write_lds(); }
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < cnt; ++k) {
        const bool more = k + 1 < cnt;
        if (more) issue(slab + (k + 1) * p.nslabs);              // next tile's global loads fly during this tile's MFMAs
        {
            const float f = da_acc_factor(Enext - Eacc);               // the running sums into this tile's unit (exact)
#pragma unroll
            for (int s = 0; s < WG_SLOTS; ++s) { acc[s][0] = acc[s][0] * f; acc[s][1] = acc[s][1] * f; }
            Eacc = Enext;
        }
        const F3 Y0 = loadY(0), Y1 = loadY(1);
        F3 F = loadF(0), Fn;
#pragma unroll
        for (int s = 0; s < WG_SLOTS; ++s) {
            if (s + 1 < WG_SLOTS) Fn = loadF(s + 1);
#pragma unroll
            for (int pr = 0; pr < 3; ++pr) {
                acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F.p[kPA[pr]], Y0.p[kPB[pr]], acc[s][0], 0, 0, 0);
                acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F.p[kPA[pr]], Y1.p[kPB[pr]], acc[s][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < WG_SLOTS) F = Fn;
        }
        if (more) {
            write_lds();
            __syncthreads();
        }
    }
    {
        const float inv1 = da_pow2(-(Eacc / 2)), inv2 = da_pow2(-(Eacc - Eacc / 2));      // back to the true unit
#pragma unroll
        for (int s = 0; s < WG_SLOTS; ++s) { acc[s][0] = acc[s][0] * inv1 * inv2; acc[s][1] = acc[s][1] * inv1 * inv2; }
    }
    // reduce the four waves' partial sums through LDS (2 rounds), then wave 0 writes this workgroup's partial dW
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(lds);
    auto put = [&](int slot) {
#pragma unroll
        for (int s = 0; s < WG_SLOTS; ++s)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) red[((slot * WG_SLOTS + s) * 2 + nn) * 64 + lane] = make_float4(acc[s][nn][0], acc[s][nn][1], acc[s][nn][2], acc[s][nn][3]);
    };
    auto add = [&](int slot) {
#pragma unroll
        for (int s = 0; s < WG_SLOTS; ++s)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) {
                const float4 v = red[((slot * WG_SLOTS + s) * 2 + nn) * 64 + lane];
                acc[s][nn][0] += v.x; acc[s][nn][1] += v.y; acc[s][nn][2] += v.z; acc[s][nn][3] += v.w;
            }
    };
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) add(wave);
    __syncthreads();
    if (wave == 1) put(0);
    __syncthreads();
    if (wave == 0) {
        add(0);
        float* part = p.partial + (size_t)blockIdx.x * p.O;
#pragma unroll
        for (int s = 0; s < WG_SLOTS; ++s)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 4 * g + reg;
                    const int tap = 2 * s + (row >> 3), ci = ch * 8 + (row & 7), co = cg * 32 + nn * 16 + i;
                    if (tap < 27 && co < p.Cout) part[((size_t)tap * p.Cin + ci) * p.Cout + co] = acc[s][nn][reg];
                }
    }
}

struct Plan { int Do, Ho, Wo, ntz, nty, ntx, ntiles; };
Plan s2n_plan(int N, int D, int H, int W) {
    Plan q;
    q.Do = (D - 1) / 2 + 1; q.Ho = (H - 1) / 2 + 1; q.Wo = (W - 1) / 2 + 1;
    q.ntz = (q.Do + TZ - 1) / TZ; q.nty = (q.Ho + TY - 1) / TY; q.ntx = (q.Wo + TX - 1) / TX;
    q.ntiles = N * q.ntz * q.nty * q.ntx;
    return q;
}
int s2n_wgrad_slabs(int ntiles, int nchunks, int ngroups) {
    int s = 256 / (nchunks * ngroups);                          // one workgroup per CU
    s = s / 8 * 8; if (s < 8) s = 8;                             // multiple of 8: the chunk workgroups of one slab share an XCD
    if (s > ntiles) s = ntiles;
    return s < 1 ? 1 : s;
}
template <typename K> int s2n_set_lds(K kern, size_t bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

// Shapes the native kernels take: one tensor input, 8-channel chunks, 32-wide output groups (the registration encoder: 16 -> 32, 32 -> 32),
// samples below 4 GiB (32-bit buffer offsets).
bool da_conv3_s2n_supported(int Cin, int Cout, int N, int D, int H, int W) {
    if (Cin % 16 != 0 || Cin > 32 || Cout % 32 != 0 || Cout > 64) return false;
    const long long in_bytes = (long long)D * H * W * Cin * 4;
    return in_bytes < (1ll << 32) && (long long)N * D * H * W < (1ll << 31);
}

size_t da_conv3_s2n_ws_bytes(int N, int D, int H, int W, int Cin, int Cout) {
    const Plan q = s2n_plan(N, D, H, W);
    const size_t pack_f = (size_t)(Cin / 8) * NSTEPS * (Cout / 16) * NPLN * 1024;
    const size_t pack_d = (size_t)27 * (Cout / 32) * (Cin / 16) * NPLN * 1024;
    const size_t part = (size_t)s2n_wgrad_slabs(q.ntiles, Cin / 8, Cout / 32) * 27 * Cin * Cout * sizeof(float);
    size_t m = pack_f > pack_d ? pack_f : pack_d;
    if (part > m) m = part;
    return da_align(m) + 256;        // (+ 256: the weight scale exponents in front of the packed operand)
}

int da_conv3_s2n_fwd(const float* in, int Cin, const float* w_tio, const float* bias, float* out,
                     int N, int D, int H, int W, int Cout, float slope, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < da_conv3_s2n_ws_bytes(N, D, H, W, Cin, Cout)) return DA_ERR_WS_SMALL;
    const Plan q = s2n_plan(N, D, H, W);
    FwdP p;
    // the packed operand: in the workspace, or in the caller's kept buffer (conv3d_internal.h: da_pp_lookup)
    const DaKeptPack kp = da_pp_lookup(w_tio, da_align((size_t)(Cin / 8) * NSTEPS * (Cout / 16) * NPLN * 1024) + 256, DA_PP_S2N_FWD);
    if (kp.only && !kp.buf) return 0;
    unsigned char* pk = kp.buf ? kp.buf : (unsigned char*)ws;
    p.in = in; p.wexp = (const int*)pk; p.wp = pk + 256; p.bias = bias; p.out = out;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Do = q.Do; p.Ho = q.Ho; p.Wo = q.Wo;
    p.ntz = q.ntz; p.nty = q.nty; p.ntx = q.ntx; p.nchunks = Cin / 8; p.NT = Cout / 16; p.slope = slope;
    if (!kp.buf || kp.fill) {
        hipLaunchKernelGGL(s2n_pack_fwd_kernel, dim3(p.nchunks, 4), dim3(256), 0, st, w_tio, (unsigned short*)(pk + 256), (int*)pk, Cin, Cout, p.NT);
        DA_LAUNCH_CHECK();
    }
    if (kp.only) return 0;
    static bool attr = false;
    if (!attr) { const int e = s2n_set_lds(s2n_fwd_kernel, NPLN * PLANE_B + 16); if (e) return e; attr = true; }
    hipLaunchKernelGGL(s2n_fwd_kernel, dim3(q.ntiles, Cout / 32), dim3(256), NPLN * PLANE_B + 16, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}

int da_conv3_s2n_dgrad(const float* dy, const float* w_tio, float* dx, int Cin, int N, int D, int H, int W, int Cout,
                       void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < da_conv3_s2n_ws_bytes(N, D, H, W, Cin, Cout)) return DA_ERR_WS_SMALL;
    const Plan q = s2n_plan(N, D, H, W);
    DgP p;
    const DaKeptPack kp = da_pp_lookup(w_tio, da_align((size_t)27 * (Cout / 32) * (Cin / 16) * NPLN * 1024) + 256, DA_PP_S2N_DGRAD);
    if (kp.only && !kp.buf) return 0;
    unsigned char* pk = kp.buf ? kp.buf : (unsigned char*)ws;
    p.dy = dy; p.wexp = (const int*)pk; p.wp = pk + 256; p.dx = dx;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Do = q.Do; p.Ho = q.Ho; p.Wo = q.Wo;
    p.ntz = q.ntz; p.nty = q.nty; p.ntx = q.ntx; p.KS = Cout / 32; p.NTN = Cin / 16;
    if (!kp.buf || kp.fill) {
        hipLaunchKernelGGL(s2n_pack_dgrad_kernel, dim3(da_grid((long long)27 * p.KS * p.NTN * 64, 256, 32)), dim3(256), 0, st, w_tio, (unsigned short*)(pk + 256), (int*)pk, Cin, Cout, p.KS, p.NTN);
        DA_LAUNCH_CHECK();
    }
    if (kp.only) return 0;
    const size_t shm = (size_t)NPLN * (Cout / 8) * GV * 16 + 16;
    static bool attr[3] = {false, false, false};
    int e = 0;
    switch (p.NTN) {
        case 1: if (!attr[1]) { e = s2n_set_lds(s2n_dgrad_kernel<1>, 98304); if (e) return e; attr[1] = true; }
                hipLaunchKernelGGL(s2n_dgrad_kernel<1>, dim3(q.ntiles), dim3(256), shm, st, p); break;
        case 2: if (!attr[2]) { e = s2n_set_lds(s2n_dgrad_kernel<2>, 98304); if (e) return e; attr[2] = true; }
                hipLaunchKernelGGL(s2n_dgrad_kernel<2>, dim3(q.ntiles), dim3(256), shm, st, p); break;
        default: return DA_ERR_UNSUPPORTED;
    }
    DA_LAUNCH_CHECK();
    return 0;
}

int da_conv3_s2n_wgrad(const float* in, int Cin, const float* dy, float* dw_tio, int N, int D, int H, int W, int Cout,
                       void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < da_conv3_s2n_ws_bytes(N, D, H, W, Cin, Cout)) return DA_ERR_WS_SMALL;
    const Plan q = s2n_plan(N, D, H, W);
    WgP p;
    p.in = in; p.dy = dy; p.partial = (float*)ws;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Do = q.Do; p.Ho = q.Ho; p.Wo = q.Wo;
    p.ntz = q.ntz; p.nty = q.nty; p.ntx = q.ntx; p.ntiles = q.ntiles;
    const int nchunks = Cin / 8, ngroups = Cout / 32;
    p.nslabs = s2n_wgrad_slabs(q.ntiles, nchunks, ngroups); p.O = 27 * Cin * Cout;
    // every (chunk, group) workgroup of a slab writes its own (tap, cin chunk, cout group) block of the slab's partial: the blocks are
    // disjoint and together cover all O entries, so the partial needs no zero fill
    const size_t shm = (size_t)NPLN * PLANE_B + NPLN * YPLANE_B + 32;
    static bool attr = false;
    if (!attr) { const int e = s2n_set_lds(s2n_wgrad_kernel, shm); if (e) return e; attr = true; }
    hipLaunchKernelGGL(s2n_wgrad_kernel, dim3(p.nslabs, nchunks, ngroups), dim3(256), shm, st, p);
    DA_LAUNCH_CHECK();
    return da_reduce_partials(p.partial, p.nslabs, p.O, dw_tio, st);
}
