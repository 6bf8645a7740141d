// Registration losses of SURVEY.md row f2 (gfx950): VoxelMorphLNCC (lib/loss.py:589-617, registry name 'lncc') and gradientLoss
// (lib/loss.py:625-671, 'gradient').  Both are HBM-bound stencil + reduction passes; reductions go through per-block double
// partials and a one-block finalize (deterministic, no float atomics).
#include "common.h"
#include <cstdlib>
#include <utility>

namespace {

constexpr int kBlocks = 4096;

__global__ void sum_partials_kernel(const double* __restrict__ partial, int count, double scale, double offset, float* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += blockDim.x) s += partial[i];
    s = da_block_sum(s, red);
    if (threadIdx.x == 0) out[0] = (float)(offset + scale * s);
}

// ------------------------------------------------------------------------------------------------
// LNCC.  The reference runs five F^3 all-ones convolutions (I, J, I^2, J^2, IJ; valid padding) and then, per window,
//   cross = S_IJ - Ibar S_J - Jbar S_I + Ibar Jbar n ;  Ivar = S_II - 2 Ibar S_I + Ibar^2 n ;  Jvar likewise ;
//   cc = cross^2 / (Ivar Jvar + eps) ;  loss = 1 - mean(cc).
// An all-ones F^3 filter is separable: three 1-D box sums (W, H, D).  Fields are planar [5][N][z][y][x] fp32.
// ------------------------------------------------------------------------------------------------
// pass 1: products + box sum along W:  t1[k][n][z][y][xo], xo < Wo = W - F + 1
__global__ void lncc_boxw_kernel(const float* __restrict__ I, const float* __restrict__ J, float* __restrict__ t1,
                                 long long rows, int W, int Wo, int F, long long plane, int dil, int stride) {
    const long long total = rows * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo); const long long r = i / Wo;
        const float* a = I + r * W + (long long)xo * stride; const float* b = J + r * W + (long long)xo * stride;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
        for (int t = 0; t < F; ++t) {
            const float x = a[t * dil], y = b[t * dil];
            s0 += x; s1 += y; s2 += x * x; s3 += y * y; s4 += x * y;
        }
        t1[i] = s0; t1[plane + i] = s1; t1[2 * plane + i] = s2; t1[3 * plane + i] = s3; t1[4 * plane + i] = s4;
    }
}

// generic 1-D box sum of K planar fields along one axis of a [M][A][B] view (axis length A, inner stride B):
// transposed == 0 (forward, valid):  out[m][ao][b] = sum_{t<F} in[m][ao*stride + t*dil][b],            ao < Ao
// transposed != 0 (its adjoint):     out[m][a][b]  = sum_{t<F, (a - t*dil) % stride == 0} in[m][(a - t*dil)/stride][b],  a < Ao
__global__ void box_axis_kernel(const float* __restrict__ in, float* __restrict__ out, long long M, int A, int Ao, long long B, int F,
                                int transposed, int dil, int stride) {
    const long long total = M * Ao * B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i % B; long long r = i / B;
        const int ao = (int)(r % Ao); const long long m = r / Ao;
        const float* src = in + (m * A) * B + b;
        float s = 0.f;
        for (int t = 0; t < F; ++t) {
            int a;
            if (!transposed) a = ao * stride + t * dil;
            else { const int rem = ao - t * dil; if (rem < 0 || rem % stride) continue; a = rem / stride; }
            if (a < A) s += src[(long long)a * B];
        }
        out[i] = s;
    }
}

__device__ __forceinline__ void lncc_terms(float sI, float sJ, float sII, float sJJ, float sIJ, float n, float eps,
                                           float& cross, float& ivar, float& jvar, float& den) {
    // same operation order as lib/loss.py:606-612 (fp32)
    const float im = sI / n, jm = sJ / n;
    cross = sIJ - im * sJ - jm * sI + im * jm * n;
    ivar = sII - 2.f * im * sI + im * im * n;
    jvar = sJJ - 2.f * jm * sJ + jm * jm * n;
    den = ivar * jvar + eps;
}

// pass 3: box sum along D of the 5 fields (in: [5][N][D][Ho*Wo]) fused with cc and its reduction; writes the window sums
// (kept for the backward pass): sums[k][n][zo][yo*xo]
__global__ void lncc_boxd_cc_kernel(const float* __restrict__ t2, float* __restrict__ sums, int N, int D, int Do, long long HW,
                                    int F, float n, float eps, double* __restrict__ partial, int dil, int stride) {
    __shared__ double red[4];
    const long long pin = (long long)N * D * HW, pout = (long long)N * Do * HW;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pout; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i % HW; long long r = i / HW;
        const int zo = (int)(r % Do); const long long nn = r / Do;
        const float* src = t2 + (nn * D + (long long)zo * stride) * HW + b;
        float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < F; ++t)
#pragma unroll
            for (int k = 0; k < 5; ++k) s[k] += src[k * pin + (long long)t * dil * HW];
#pragma unroll
        for (int k = 0; k < 5; ++k) sums[k * pout + i] = s[k];
        float cross, ivar, jvar, den;
        lncc_terms(s[0], s[1], s[2], s[3], s[4], n, eps, cross, ivar, jvar, den);
        acc += (double)((cross * cross) / den);
    }
    const double tot = da_block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// backward, per window: A = d cc / d cross, B = d cc / d Ivar, C = d cc / d Jvar (times -dloss / M), and the fields whose
// transposed box sums give the voxel gradients:
//   dL/dI_p = J_p [A] - [A Jbar] + 2 I_p [B] - 2 [B Ibar] ;  dL/dJ_p = I_p [A] - [A Ibar] + 2 J_p [C] - 2 [C Jbar]
// ([.] = sum over the windows containing p).  G: [7][N][Do][Ho][Wo] = A, A Ibar, A Jbar, B, B Ibar, C, C Jbar
__global__ void lncc_bwd_fields_kernel(const float* __restrict__ sums, float* __restrict__ G, long long P, float n, float eps,
                                       const float* __restrict__ dloss, float inv_m) {
    const float gs = -dloss[0] * inv_m;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
        const float sI = sums[i], sJ = sums[P + i];
        float cross, ivar, jvar, den;
        lncc_terms(sI, sJ, sums[2 * P + i], sums[3 * P + i], sums[4 * P + i], n, eps, cross, ivar, jvar, den);
        const float im = sI / n, jm = sJ / n;
        const float A = gs * 2.f * cross / den;
        const float q = gs * -(cross * cross) / (den * den);
        const float B = q * jvar, C = q * ivar;
        G[i] = A; G[P + i] = A * im; G[2 * P + i] = A * jm; G[3 * P + i] = B; G[4 * P + i] = B * im; G[5 * P + i] = C; G[6 * P + i] = C * jm;
    }
}

// last backward pass: full box sum along W of the 7 fields (in: [7][rows][Wo]) fused with the combine
__global__ void lncc_bwd_boxw_combine_kernel(const float* __restrict__ t, const float* __restrict__ I, const float* __restrict__ J,
                                             float* __restrict__ dI, float* __restrict__ dJ, long long rows, int W, int Wo, int F,
                                             int dil, int stride) {
    const long long total = rows * W, plane = rows * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); const long long r = i / W;
        float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int tt = 0; tt < F; ++tt) {
            const int rem = x - tt * dil;
            if (rem < 0 || rem % stride) continue;
            const int xo = rem / stride;
            if (xo < Wo) {
#pragma unroll
                for (int k = 0; k < 7; ++k) s[k] += t[k * plane + r * Wo + xo];
            }
        }
        const float a = I[i], b = J[i];
        if (dI) dI[i] = b * s[0] - s[2] + 2.f * a * s[3] - 2.f * s[4];
        if (dJ) dJ[i] = a * s[0] - s[1] + 2.f * b * s[5] - 2.f * s[6];
    }
}

// ------------------------------------------------------------------------------------------------
// LNCC, marching form (round 5; dilation 1, stride 1, F = 5 or 9 -- VoxelMorphLNCC as the registry builds it).  The separable passes
// above move the five (seven) intermediate fields through HBM three times; here a workgroup owns a TW x TH tile of windows and walks
// z.  Per plane: every thread loads its share of the (TH+F-1) x (TW+F-1) halo tile, forms the five fields and puts them into LDS; the
// x pass reads 8+F-1 consecutive values per (field, row, segment) task and writes eight sums back; the y pass gives each thread two
// windows of one column; the last F plane sums of those stay in registers (the z loop is unrolled F times so that the ring slot is a
// compile-time index) and are re-added every plane -- no running subtraction, so nothing drifts along z.  HBM sees I and J once
// (halo re-reads are L2 hits) plus, in the forward pass, the five window sums the backward pass starts from.
// The backward pass is the same walk over the zero-padded WINDOW fields (the adjoint of a valid box sum is a full one):
//   dL/dI_p = J_p [A] + 2 I_p [B] - [A Jbar + 2 B Ibar] ,  dL/dJ_p = I_p [A] + 2 J_p [C] - [A Ibar + 2 C Jbar]
// -- five fields again, formed from the five window sums when a plane is staged.
// ------------------------------------------------------------------------------------------------
template <int... Is, class Fn> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, Fn&& fn) { (fn(std::integral_constant<int, Is>{}), ...); }
template <int N, class Fn> __device__ __forceinline__ void static_for(Fn&& fn) { static_for_impl(std::make_integer_sequence<int, N>{}, fn); }

typedef float f2 __attribute__((ext_vector_type(2)));
// N eight-byte LDS reads at byte address a + k * STEP, each a plain ds_read_b64; the caller waits for lgkmcnt
template <int K, int N, int STEP> __device__ __forceinline__ void lds_col_b64(f2* c, unsigned a) {
    if constexpr (K < N) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(c[K]) : "v"(a), "n"(K * STEP));
        lds_col_b64<K + 1, N, STEP>(c, a);
    }
}

struct MarchP {
    const float* I; const float* J; const float* sums_in; float* sums_out; float* dI; float* dJ; double* partial; const float* dloss;
    int N, Liz, Liy, Lix, Loz, Loy, Lox, pad, ZC, ntx, nty, nch, nwg;
    long long P;                       // windows per field (N * Do * Ho * Wo)
    float n, eps, inv_m;
};

// What the forward pass leaves per window for the backward pass (the marching form's `sums`; the separable form keeps the five raw sums):
//   A' = 2 cross / den,  B' = -cross^2 Jvar / den^2,  C' = -cross^2 Ivar / den^2,  E1' = A' Jbar + 2 B' Ibar,  E2' = A' Ibar + 2 C' Jbar
// -- the backward fields up to the factor -dloss / M, which is applied to the box sums at the end (they are linear): the backward staging does no
// arithmetic at all.
// The five fields travel as three PAIRS -- forward (I, J), (I^2, J^2), (I J, -); backward (A', B'), (C', E1'), (E2', -) -- one v_pk_add_f32 per
// add of two fields.  LDS rows are padded so that the lanes of one 16-byte access group touch different bank groups.  All global accesses are
// buffer instructions with per-thread offsets fixed before the walk (an out-of-tile or out-of-volume position is offset 0xFFFFFFFF: loads
// return 0, stores are dropped) and per-plane descriptors built on the SALU.
// The kernel is bound by latency, not by a pipe (ablations: profiles/r05_lncc_marching.txt), so the three stages of three consecutive planes
// share ONE barrier interval: y pass + z ring + epilogue of plane i, x pass of plane i + 1 (xs double-buffered), staging of plane i + 2 (the
// forward pass double-buffers its 8 KB (I, J) tile too; the backward pass, whose tile is three pairs, stages after a second barrier).
template <int F, bool BWD>
__global__ void __launch_bounds__(256, 2) lncc_march_kernel(MarchP p) {
    static_assert(F == 9 || F == 5, "x-pass reads whole 16-byte pieces; operand lists of the LDS waits");
    constexpr int G = 3, TW = 32, TH = 16, IW = TW + F - 1, IH = TH + F - 1, NPOS = IH * IW, NSL = (NPOS + 255) / 256, NIN = BWD ? 5 : 2;
    constexpr int SEG = 16, NSEG = TW / SEG, NTASK = G * IH * NSEG, RD = SEG + F - 1;
    static_assert(IH * NSEG <= 64, "one wave per field pair in the x pass");
    constexpr int RS = IW + 2, XS = TW + 2;                     // row strides in pairs: (stride * 8 bytes / 16) odd
    constexpr unsigned OOB = 0xFFFFFFFFu;
    constexpr int GR = BWD ? G : 1;                             // forward: only (I, J) is staged, the x pass forms the products of its group
    constexpr int RB = BWD ? 1 : 2;                             // staging buffers
    constexpr int RROWS = (NSL * 256 + IW - 1) / IW;            // staging rows incl. the spare ones the last (partly out-of-tile) slot writes zeros to
    constexpr int RAWN = GR * RROWS * RS, XSN = G * IH * XS;
    __shared__ __attribute__((aligned(16))) f2 raw[RB * RAWN];
    __shared__ __attribute__((aligned(16))) f2 xs[2 * XSN];
    __shared__ double red[4];
    const int t = threadIdx.x;
    // workgroup -> tile: x fastest, then y, then z chunk, then sample; consecutive ids go round the eight XCDs (giving each XCD a contiguous run
    // of tiles, for the shared halos, measured slower: 87 vs 75 us backward)
    const int wid = blockIdx.x;
    const int tile = wid % (p.ntx * p.nty), chunk = (wid / (p.ntx * p.nty)) % p.nch, n = wid / (p.ntx * p.nty * p.nch);
    const int x0 = (tile % p.ntx) * TW, y0 = (tile / p.ntx) * TH;
    const int zo0 = chunk * p.ZC, zo1 = min(zo0 + p.ZC, p.Loz);
    const int nplanes = zo1 - zo0 + F - 1;
    const long long plane_in = (long long)p.Liy * p.Lix, plane_out = (long long)p.Loy * p.Lox;
    const unsigned in_bytes = (unsigned)(plane_in * 4), out_bytes = (unsigned)(plane_out * 4);
    auto rsrc = [](const float* ptr, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, bytes, 0x00020000); };
    auto ld = [](__amdgpu_buffer_rsrc_t r, unsigned o) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0)); };
    auto st = [](__amdgpu_buffer_rsrc_t r, unsigned o, float x) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r, o, 0, 0); };
    unsigned voff[NSL]; int lofs[NSL];
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        const int idx = t + 256 * j, iy = idx / IW, ix = idx % IW;
        const int gy = y0 - p.pad + iy, gx = x0 - p.pad + ix;
        const bool ok = idx < NPOS && gy >= 0 && gy < p.Liy && gx >= 0 && gx < p.Lix;
        voff[j] = ok ? (unsigned)(gy * p.Lix + gx) * 4u : OOB;
        lofs[j] = iy * RS + ix;
    }
    const int lx = t % TW, ys = t / TW;
    unsigned ooff[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int yy = y0 + 2 * ys + o, xx = x0 + lx;
        ooff[o] = (yy < p.Loy && xx < p.Lox) ? (unsigned)(yy * p.Lox + xx) * 4u : OOB;
    }
    // descriptors span one sample of one field and are built once; a plane is a scalar byte offset (the launcher keeps a sample below 2 GiB); a
    // plane outside the volume / the chunk keeps its offset inside the sample and loads (stores) through the EMPTY descriptor instead
    const unsigned in_sample = in_bytes * (unsigned)p.Liz, out_sample = out_bytes * (unsigned)p.Loz;
    const __amdgpu_buffer_rsrc_t r_null = rsrc(p.I, 0u);
    const float* in0 = BWD ? p.sums_in : p.I;
    const float* in1 = BWD ? p.sums_in + p.P : p.J;
    const __amdgpu_buffer_rsrc_t r_in0 = rsrc(in0 + (long long)n * p.Liz * plane_in, in_sample), r_in1 = rsrc(in1 + (long long)n * p.Liz * plane_in, in_sample);
    const __amdgpu_buffer_rsrc_t r_in2 = rsrc(BWD ? p.sums_in + 2 * p.P + (long long)n * p.Liz * plane_in : p.I, BWD ? in_sample : 0u);
    const __amdgpu_buffer_rsrc_t r_in3 = rsrc(BWD ? p.sums_in + 3 * p.P + (long long)n * p.Liz * plane_in : p.I, BWD ? in_sample : 0u);
    const __amdgpu_buffer_rsrc_t r_in4 = rsrc(BWD ? p.sums_in + 4 * p.P + (long long)n * p.Liz * plane_in : p.I, BWD ? in_sample : 0u);
    auto ldo = [](__amdgpu_buffer_rsrc_t r, unsigned o, unsigned so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o, so, 0)); };
    auto sto = [](__amdgpu_buffer_rsrc_t r, unsigned o, unsigned so, float x) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r, o, so, 0); };
    float v[NIN][NSL];
    auto issue = [&](int i) {                                   // input plane i of the chunk (zeros outside the volume / past the chunk)
        const int zin = zo0 - p.pad + i;
        const bool zok = i < nplanes && zin >= 0 && zin < p.Liz;
        const unsigned so = zok ? (unsigned)zin * in_bytes : 0u;
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            v[0][j] = ldo(zok ? r_in0 : r_null, voff[j], so); v[1][j] = ldo(zok ? r_in1 : r_null, voff[j], so);
            if constexpr (BWD) { v[2][j] = ldo(zok ? r_in2 : r_null, voff[j], so); v[3][j] = ldo(zok ? r_in3 : r_null, voff[j], so); v[4][j] = ldo(zok ? r_in4 : r_null, voff[j], so); }
        }
    };
    const __amdgpu_buffer_rsrc_t r_pi = rsrc(p.I + (long long)n * p.Loz * plane_out, BWD ? out_sample : 0u), r_pj = rsrc(p.J + (long long)n * p.Loz * plane_out, BWD ? out_sample : 0u);
    float pi[2] = {0.f, 0.f}, pj[2] = {0.f, 0.f};               // backward: I, J at this thread's two voxels of the output plane of walk step i
    auto issue_post = [&](int i) {
        if constexpr (BWD) {
            const int zo = zo0 + i - (F - 1);
            const bool zok = zo >= zo0 && zo < zo1;
            const unsigned so = zok ? (unsigned)zo * out_bytes : 0u;
#pragma unroll
            for (int o = 0; o < 2; ++o) { pi[o] = ldo(zok ? r_pi : r_null, ooff[o], so); pj[o] = ldo(zok ? r_pj : r_null, ooff[o], so); }
        }
    };
    // outputs: backward dI, dJ; forward the five window terms
    const float* o0 = BWD ? p.dI : p.sums_out;
    const float* o1 = BWD ? p.dJ : p.sums_out + p.P;
    const long long osmp = (long long)n * p.Loz * plane_out;
    const __amdgpu_buffer_rsrc_t r_o0 = rsrc(o0 ? o0 + osmp : nullptr, o0 ? out_sample : 0u), r_o1 = rsrc(o1 ? o1 + osmp : nullptr, o1 ? out_sample : 0u);
    const __amdgpu_buffer_rsrc_t r_o2 = rsrc(BWD ? nullptr : p.sums_out + 2 * p.P + osmp, BWD ? 0u : out_sample);
    const __amdgpu_buffer_rsrc_t r_o3 = rsrc(BWD ? nullptr : p.sums_out + 3 * p.P + osmp, BWD ? 0u : out_sample);
    const __amdgpu_buffer_rsrc_t r_o4 = rsrc(BWD ? nullptr : p.sums_out + 4 * p.P + osmp, BWD ? 0u : out_sample);
    float gs = 0.f;
    if constexpr (BWD) gs = -p.dloss[0] * p.inv_m;
    const float inv_n = 1.f / p.n;
    auto stage = [&](int rb) {
        f2* dst = raw + rb * RAWN;
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            dst[lofs[j]] = (f2){v[0][j], v[1][j]};
            if constexpr (BWD) { dst[RROWS * RS + lofs[j]] = (f2){v[2][j], v[3][j]}; dst[2 * RROWS * RS + lofs[j]] = (f2){v[4][j], v[4][j]}; }
        }
    };
    // x pass, one task per thread (the first NTASK threads): lanes run over rows first (conflict-free 16-byte LDS accesses with the padded
    // strides).  Two sums of a segment are added up (outputs 0 and 8); the others slide from them (+ entering - leaving value), two independent
    // chains of seven.
    const int xg = __builtin_amdgcn_readfirstlane(t >> 6), xl = t & 63;      // wave g does group g: the product branch is wave-uniform
    const bool xact = xg < G && xl < IH * NSEG;
    const int xrow = xl % IH, xseg = xl / IH;
    const int xsrc = (BWD ? xg * (RROWS * RS) : 0) + xrow * RS + xseg * SEG, xdst = xg * (IH * XS) + xrow * XS + xseg * SEG;
    auto xtask = [&](auto gc, int rb, int xb) {                 // one instantiation per pair: no values merge between the product forms
        constexpr int g = decltype(gc)::value;
        const f2* src = raw + rb * RAWN + xsrc;
        f2 pv[RD];
#pragma unroll
        for (int k = 0; k < RD; ++k) pv[k] = src[k];
        if constexpr (!BWD && g == 1) {
#pragma unroll
            for (int k = 0; k < RD; ++k) pv[k] = pv[k] * pv[k];
        }
        if constexpr (!BWD && g == 2) {
#pragma unroll
            for (int k = 0; k < RD; ++k) pv[k].x = pv[k].x * pv[k].y;                  // .y of the third pair is never read
        }
        f2 so[SEG];
        auto anchor = [&](int o) {
            if constexpr (F == 9) return ((pv[o] + pv[o + 1]) + pv[o + 2]) + ((pv[o + 3] + pv[o + 4]) + pv[o + 5]) + ((pv[o + 6] + pv[o + 7]) + pv[o + 8]);
            else return ((pv[o] + pv[o + 1]) + (pv[o + 2] + pv[o + 3])) + pv[o + 4];
        };
        so[0] = anchor(0); so[SEG / 2] = anchor(SEG / 2);
#pragma unroll
        for (int o = 1; o < SEG / 2; ++o) {
            so[o] = so[o - 1] + (pv[o + F - 1] - pv[o - 1]);
            so[SEG / 2 + o] = so[SEG / 2 + o - 1] + (pv[SEG / 2 + o + F - 1] - pv[SEG / 2 + o - 1]);
        }
        f2* dst = xs + xb * XSN + xdst;
#pragma unroll
        for (int o = 0; o < SEG; ++o) dst[o] = so[o];
    };
    auto xpass = [&](int rb, int xb) {
        if (xact) {
            if (xg == 0) xtask(std::integral_constant<int, 0>{}, rb, xb);
            else if (xg == 1) xtask(std::integral_constant<int, 1>{}, rb, xb);
            else xtask(std::integral_constant<int, 2>{}, rb, xb);
        }
    };
    // y pass column reads as plain ds_read_b64 (256 B/clk): left to itself the compiler pairs them into ds_read2_b64, which runs at half that
    const unsigned ybase = (unsigned)(((2 * ys) * XS + lx) * 8) + (unsigned)(size_t)(&xs[0]);
    auto wait_cols = [&](f2* c) {                               // the wait carries the read registers as operands: nothing may use them before it
        if constexpr (F == 9) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]) :: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) :: "memory");
    };
    f2 P[G][2][F], T3[G][2][F];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int k = 0; k < F; ++k) { P[g][o][k] = (f2){0.f, 0.f}; T3[g][o][k] = (f2){0.f, 0.f}; }
    double acc = 0.0;
    // prologue: plane 0 through the x pass, plane 1 staged, plane 2 on its way
    issue(0);
    stage(0);
    issue(1);
    __syncthreads();
    xpass(0, 0);
    if constexpr (RB == 1) __syncthreads();
    stage(RB == 2 ? 1 : 0);
    issue(2);
    issue_post(0);
    __syncthreads();
#pragma unroll 1
    for (int ib = 0; ib < nplanes; ib += F) {
        static_for<F>([&](auto rc) {
            constexpr int r = decltype(rc)::value;                   // the ring slot, a compile-time index
            const int i = ib + r;                                    // walk steps past the chunk (the walk is padded to whole rings: a conditional slot would
            const int par = i & 1;                                   // keep all 2 x F ring registers of every field alive) load zeros and store nothing
            float ci[2], cj[2];
            if constexpr (BWD) { ci[0] = pi[0]; ci[1] = pi[1]; cj[0] = pj[0]; cj[1] = pj[1]; }
            issue_post(i + 1);
            xpass(RB == 2 ? par ^ 1 : 0, par ^ 1);
            if constexpr (RB == 2) { stage(par); issue(i + 3); }
            f2 c[G][F + 1];
#pragma unroll
            for (int g = 0; g < G; ++g) lds_col_b64<0, F + 1, XS * 8>(c[g], ybase + (unsigned)((par * XSN + g * IH * XS) * 8));
#pragma unroll
            for (int g = 0; g < G; ++g) wait_cols(c[g]);
            f2 s[G][2];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                f2 m;
                if constexpr (F == 9) m = ((c[g][1] + c[g][2]) + (c[g][3] + c[g][4])) + ((c[g][5] + c[g][6]) + (c[g][7] + c[g][8]));
                else m = (c[g][1] + c[g][2]) + (c[g][3] + c[g][4]);
                P[g][0][r] = m + c[g][0]; P[g][1][r] = m + c[g][F];
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    if constexpr (F == 9) {
                        T3[g][o][r] = P[g][o][r] + P[g][o][(r + F - 1) % F] + P[g][o][(r + F - 2) % F];
                        s[g][o] = T3[g][o][r] + T3[g][o][(r + F - 3) % F] + T3[g][o][(r + F - 6) % F];
                    } else {
                        s[g][o] = ((P[g][o][0] + P[g][o][1]) + (P[g][o][2] + P[g][o][3])) + P[g][o][4];
                    }
                }
            }
            if (i >= F - 1) {
                const int zo = zo0 + i - (F - 1);
                const bool zin = zo < zo1;
                const unsigned so = zin ? (unsigned)zo * out_bytes : 0u;
                if constexpr (BWD) {
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        sto(zin ? r_o0 : r_null, ooff[o], so, gs * (cj[o] * s[0][o].x + 2.f * ci[o] * s[0][o].y - s[1][o].y));
                        sto(zin ? r_o1 : r_null, ooff[o], so, gs * (ci[o] * s[0][o].x + 2.f * cj[o] * s[1][o].x - s[2][o].x));
                    }
                } else {
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        const float sI = s[0][o].x, sJ = s[0][o].y, sII = s[1][o].x, sJJ = s[1][o].y, sIJ = s[2][o].x;
                        const float im = sI * inv_n, jm = sJ * inv_n;
                        const float cross = sIJ - im * sJ - jm * sI + im * jm * p.n;
                        const float ivar = sII - 2.f * im * sI + im * im * p.n, jvar = sJJ - 2.f * jm * sJ + jm * jm * p.n;
                        const float rd = __builtin_amdgcn_rcpf(ivar * jvar + p.eps), cr = cross * rd, q = -(cr * cr);
                        if (zin && ooff[o] != OOB) acc += (double)(cross * cr);
                        const float A = 2.f * cr, B = q * jvar, C = q * ivar;
                        sto(zin ? r_o0 : r_null, ooff[o], so, A); sto(zin ? r_o1 : r_null, ooff[o], so, B); sto(zin ? r_o2 : r_null, ooff[o], so, C);
                        sto(zin ? r_o3 : r_null, ooff[o], so, A * jm + 2.f * B * im); sto(zin ? r_o4 : r_null, ooff[o], so, A * im + 2.f * C * jm);
                    }
                }
            }
            __syncthreads();
            if constexpr (RB == 1) { stage(0); issue(i + 3); __syncthreads(); }
        });
    }
    if constexpr (!BWD) {
        const double tot = da_block_sum(acc, red);
        if (t == 0) p.partial[wid] = tot;
    }
}

struct MarchGeom { int ntx, nty, nch, ZC; bool ok; };
// Which form a call takes is decided ONCE, from the INPUT extents (D, H, W) -- da_lncc_ws_bytes, da_lncc_fwd and da_lncc_bwd must agree: the forward
// writes the marching form's five backward terms into `sums` where the separable form keeps raw window sums, and the workspace is sized per form.
// (Both walks address at most (D + F)(H + F)(W + F) elements per sample with 32-bit byte offsets.)  The tiling below is per walk direction.
static bool lncc_march_ok(int D, int H, int W, int F, int dil, int stride) {
    static const int off = [] { const char* e = getenv("DA_LNCC_MARCH"); return (e && e[0] == '0') ? 1 : 0; }();
    return !off && dil == 1 && stride == 1 && (F == 9 || F == 5) && (long long)(D + F) * (H + F) * (W + F) * 4 < (1ll << 31);
}
static MarchGeom lncc_march_geom(int N, int Lz, int Ly, int Lx, int F, bool ok) {
    MarchGeom g{};
    g.ok = ok;
    if (!g.ok) return g;
    g.ntx = (Lx + 31) / 32; g.nty = (Ly + 15) / 16;
    const long long tiles = (long long)g.ntx * g.nty * N;
    // z chunks: two workgroups per CU are resident (registers), so the time is (rounds of 512 workgroups) x (planes a workgroup walks: its chunk
    // + F - 1 re-read planes, padded to whole rings of F); among equal times the fewest total planes
    long long best_t = -1, best_w = -1;
    for (int nch = 1; nch <= (Lz + F - 1) / F; ++nch) {
        int ZC = (Lz + nch - 1) / nch;
        ZC = ((ZC - 1 + F - 1) / F) * F + 1;                    // ZC + F - 1 a multiple of F wastes no ring slot
        const int n2 = (Lz + ZC - 1) / ZC;
        const long long walk = ZC + F - 1, nwg = tiles * n2, t = ((nwg + 511) / 512) * walk, w = nwg * walk;
        if (best_t < 0 || t < best_t || (t == best_t && w < best_w)) { best_t = t; best_w = w; g.ZC = ZC; g.nch = n2; }
    }
    return g;
}

// ------------------------------------------------------------------------------------------------
// gradientLoss (lib/loss.py:625-671) on disp [N][D][H][W][3] (NDHWC).  Central differences WITHOUT the 1/2h and with the
// reference's sign quirk: along D it is u(+1) - u(-1), along H and W it is u(+1) + u(-1) (lib/loss.py:659-663).
// L2: mean over the axis-interior voxels of d^2, times (spatial_dims[c] spacing[c] / spacing[k])^2 where the 3-vector is
// broadcast over the CHANNEL axis (as in BendingEnergyLoss), then mean over (N, 3), then the three axes averaged.
// L1: plain mean of |d| (the weights only appear in the L2 branch).
// ------------------------------------------------------------------------------------------------
struct GradK { float k[3][3]; int l2; };     // k[c][axis]

static GradK gradloss_coeffs(int N, int D, int H, int W, const float* spacing3, int normalize, int norm) {
    float sp[3] = {1.f, 1.f, 1.f};
    if (spacing3) { sp[0] = spacing3[0]; sp[1] = spacing3[1]; sp[2] = spacing3[2]; }
    if (normalize) { const float m = fminf(sp[0], fminf(sp[1], sp[2])); sp[0] /= m; sp[1] /= m; sp[2] /= m; }
    float dims[3] = {(float)D, (float)H, (float)W};
    if (normalize) { const float m = fminf(dims[0], fminf(dims[1], dims[2])); dims[0] /= m; dims[1] /= m; dims[2] /= m; }
    const double cnt[3] = {(double)(D - 2) * H * W, (double)D * (H - 2) * W, (double)D * H * (W - 2)};
    GradK K; K.l2 = (norm == 2);
    for (int c = 0; c < 3; ++c)
        for (int k = 0; k < 3; ++k) {
            const float w = K.l2 ? dims[c] * sp[c] / sp[k] : 1.f;
            K.k[c][k] = (float)((double)(w * w) / (cnt[k] * 3.0 * (double)N * 3.0));
        }
    return K;
}

#define GU(dd, hh, ww) u[((((long long)(dd)) * H + (hh)) * W + (ww)) * 3 + c]

__global__ void gradloss_partial_kernel(const float* __restrict__ disp, int D, int H, int W, GradK K, double* __restrict__ partial) {
    __shared__ double red[4];
    const int n = blockIdx.y;
    const float* u = disp + (long long)n * D * H * W * 3;
    const long long total = (long long)D * H * W * 3;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3); long long r = i / 3;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const int d = (int)(r / H);
        float v = 0.f;
        if (d >= 1 && d < D - 1) { const float t = fabsf(GU(d + 1, h, w) - GU(d - 1, h, w)); v += K.k[c][0] * (K.l2 ? t * t : t); }
        if (h >= 1 && h < H - 1) { const float t = fabsf(GU(d, h + 1, w) + GU(d, h - 1, w)); v += K.k[c][1] * (K.l2 ? t * t : t); }
        if (w >= 1 && w < W - 1) { const float t = fabsf(GU(d, h, w + 1) + GU(d, h, w - 1)); v += K.k[c][2] * (K.l2 ? t * t : t); }
        acc += (double)v;
    }
    const double tot = da_block_sum(acc, red);
    if (threadIdx.x == 0) partial[(size_t)n * gridDim.x + blockIdx.x] = tot;
}

// gather form: voxel p gets a term from the stencil centred one step below it and one step above it, per axis
__global__ void gradloss_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ dloss, float* __restrict__ d_disp,
                                    int D, int H, int W, GradK K) {
    const int n = blockIdx.y;
    const float* u = disp + (long long)n * D * H * W * 3;
    float* du = d_disp + (long long)n * D * H * W * 3;
    const long long total = (long long)D * H * W * 3;
    const float gl = dloss[0];
    auto fp = [&](float t) -> float { return K.l2 ? 2.f * t : (t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f)); };   // d f(|t|) / dt
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3); long long r = i / 3;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const int d = (int)(r / H);
        float g = 0.f;
        // axis D: t(q) = u(q+1) - u(q-1); p = q+1 -> +f'(t), p = q-1 -> -f'(t)
        if (d - 1 >= 1 && d - 1 < D - 1) g += K.k[c][0] * fp(GU(d, h, w) - GU(d - 2, h, w));
        if (d + 1 >= 1 && d + 1 < D - 1) g -= K.k[c][0] * fp(GU(d + 2, h, w) - GU(d, h, w));
        // axes H, W: t(q) = u(q+1) + u(q-1); both neighbours get +f'(t)
        if (h - 1 >= 1 && h - 1 < H - 1) g += K.k[c][1] * fp(GU(d, h, w) + GU(d, h - 2, w));
        if (h + 1 >= 1 && h + 1 < H - 1) g += K.k[c][1] * fp(GU(d, h + 2, w) + GU(d, h, w));
        if (w - 1 >= 1 && w - 1 < W - 1) g += K.k[c][2] * fp(GU(d, h, w) + GU(d, h, w - 2));
        if (w + 1 >= 1 && w + 1 < W - 1) g += K.k[c][2] * fp(GU(d, h, w + 2) + GU(d, h, w));
        du[i] = gl * g;
    }
}

}  // namespace

// ================================================================================================
static inline int lncc_out(int L, int F, int dil, int stride) { const int span = dil * (F - 1) + 1; return L < span ? 0 : (L - span) / stride + 1; }

extern "C" size_t da_lncc_ws_bytes(int N, int D, int H, int W, int F, int dil, int stride) {
    if (F < 1 || dil < 1 || stride < 1) return 0;
    const size_t Wo = lncc_out(W, F, dil, stride), Ho = lncc_out(H, F, dil, stride), Do = lncc_out(D, F, dil, stride);
    if (!Wo || !Ho || !Do) return 0;
    const MarchGeom mg = lncc_march_geom(N, (int)Do, (int)Ho, (int)Wo, F, lncc_march_ok(D, H, W, F, dil, stride));
    if (mg.ok) return da_align((size_t)mg.ntx * mg.nty * mg.nch * N * sizeof(double));     // marching form: the loss partials only
    // forward: t1 [5][N][D][H][Wo], t2 [5][N][D][Ho][Wo], partials; backward: G [7][N][Do][Ho][Wo], [7][N][D][Ho][Wo], [7][N][D][H][Wo]
    const size_t fwd = da_align((size_t)5 * N * D * H * Wo * 4) + da_align((size_t)5 * N * D * Ho * Wo * 4) + da_align((size_t)kBlocks * 8);
    const size_t bwd = da_align((size_t)7 * N * Do * Ho * Wo * 4) + da_align((size_t)7 * N * D * Ho * Wo * 4) + da_align((size_t)7 * N * D * H * Wo * 4);
    return fwd > bwd ? fwd : bwd;
}

extern "C" int da_lncc_fwd(const float* I, const float* J, int N, int D, int H, int W, int F, int dil, int stride, float eps,
                           float* loss, float* sums, void* ws, size_t ws_bytes, void* stream) {
    if (!I || !J || !loss || !sums || N <= 0 || F < 1 || dil < 1 || stride < 1) return DA_ERR_BADARG;
    const int Wo = lncc_out(W, F, dil, stride), Ho = lncc_out(H, F, dil, stride), Do = lncc_out(D, F, dil, stride);
    if (!Wo || !Ho || !Do) return DA_ERR_BADARG;
    if (ws_bytes < da_lncc_ws_bytes(N, D, H, W, F, dil, stride)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    const MarchGeom mg = lncc_march_geom(N, Do, Ho, Wo, F, lncc_march_ok(D, H, W, F, dil, stride));
    if (mg.ok) {
        MarchP p{};
        p.I = I; p.J = J; p.sums_out = sums; p.partial = (double*)ws;
        p.N = N; p.Liz = D; p.Liy = H; p.Lix = W; p.Loz = Do; p.Loy = Ho; p.Lox = Wo; p.pad = 0; p.ZC = mg.ZC; p.ntx = mg.ntx; p.nty = mg.nty; p.nch = mg.nch; p.nwg = mg.ntx * mg.nty * mg.nch * N;
        p.P = (long long)N * Do * Ho * Wo; p.n = (float)((double)F * F * F); p.eps = eps;
        const dim3 grid((unsigned)p.nwg);
        if (F == 9) hipLaunchKernelGGL((lncc_march_kernel<9, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((lncc_march_kernel<5, false>), grid, dim3(256), 0, st, p);
        DA_LAUNCH_CHECK();
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, p.nwg, -1.0 / (double)p.P, 1.0, loss);
        DA_LAUNCH_CHECK();
        return 0;
    }
    float* t1 = (float*)ws;
    float* t2 = (float*)((char*)ws + da_align((size_t)5 * N * D * H * Wo * 4));
    double* partial = (double*)((char*)t2 + da_align((size_t)5 * N * D * Ho * Wo * 4));
    const long long rows = (long long)N * D * H;
    hipLaunchKernelGGL(lncc_boxw_kernel, dim3(da_grid(rows * Wo, 256)), dim3(256), 0, st, I, J, t1, rows, W, Wo, F, rows * Wo, dil, stride);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(box_axis_kernel, dim3(da_grid((long long)5 * N * D * Ho * Wo, 256)), dim3(256), 0, st, t1, t2, (long long)5 * N * D, H, Ho, (long long)Wo, F, 0, dil, stride);
    DA_LAUNCH_CHECK();
    const long long pout = (long long)N * Do * Ho * Wo;
    int nblocks = (int)da_cdiv(pout, 256); if (nblocks > kBlocks) nblocks = kBlocks;
    hipLaunchKernelGGL(lncc_boxd_cc_kernel, dim3(nblocks), dim3(256), 0, st, t2, sums, N, D, Do, (long long)Ho * Wo, F, (float)((double)F * F * F), eps, partial, dil, stride);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, nblocks, -1.0 / (double)pout, 1.0, loss);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_lncc_bwd(const float* I, const float* J, const float* sums, const float* dloss, float* dI, float* dJ,
                           int N, int D, int H, int W, int F, int dil, int stride, float eps, void* ws, size_t ws_bytes, void* stream) {
    if (!I || !J || !sums || !dloss || N <= 0 || F < 1 || dil < 1 || stride < 1) return DA_ERR_BADARG;
    if (!dI && !dJ) return 0;
    const int Wo = lncc_out(W, F, dil, stride), Ho = lncc_out(H, F, dil, stride), Do = lncc_out(D, F, dil, stride);
    if (!Wo || !Ho || !Do) return DA_ERR_BADARG;
    if (ws_bytes < da_lncc_ws_bytes(N, D, H, W, F, dil, stride)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    const MarchGeom mg = lncc_march_geom(N, D, H, W, F, lncc_march_ok(D, H, W, F, dil, stride));
    if (mg.ok) {
        MarchP p{};
        p.I = I; p.J = J; p.sums_in = sums; p.dI = dI; p.dJ = dJ; p.dloss = dloss;
        p.N = N; p.Liz = Do; p.Liy = Ho; p.Lix = Wo; p.Loz = D; p.Loy = H; p.Lox = W; p.pad = F - 1; p.ZC = mg.ZC; p.ntx = mg.ntx; p.nty = mg.nty; p.nch = mg.nch; p.nwg = mg.ntx * mg.nty * mg.nch * N;
        p.P = (long long)N * Do * Ho * Wo; p.n = (float)((double)F * F * F); p.eps = eps; p.inv_m = (float)(1.0 / (double)p.P);
        const dim3 grid((unsigned)p.nwg);
        if (F == 9) hipLaunchKernelGGL((lncc_march_kernel<9, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((lncc_march_kernel<5, true>), grid, dim3(256), 0, st, p);
        DA_LAUNCH_CHECK();
        return 0;
    }
    float* G = (float*)ws;
    float* g1 = (float*)((char*)ws + da_align((size_t)7 * N * Do * Ho * Wo * 4));
    float* g2 = (float*)((char*)g1 + da_align((size_t)7 * N * D * Ho * Wo * 4));
    const long long P = (long long)N * Do * Ho * Wo;
    hipLaunchKernelGGL(lncc_bwd_fields_kernel, dim3(da_grid(P, 256)), dim3(256), 0, st, sums, G, P, (float)((double)F * F * F), eps, dloss, (float)(1.0 / (double)P));
    DA_LAUNCH_CHECK();
    // adjoint of the box filter along D, then H, then W (+ combine)
    hipLaunchKernelGGL(box_axis_kernel, dim3(da_grid((long long)7 * N * D * Ho * Wo, 256)), dim3(256), 0, st, G, g1, (long long)7 * N, Do, D, (long long)Ho * Wo, F, 1, dil, stride);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(box_axis_kernel, dim3(da_grid((long long)7 * N * D * H * Wo, 256)), dim3(256), 0, st, g1, g2, (long long)7 * N * D, Ho, H, (long long)Wo, F, 1, dil, stride);
    DA_LAUNCH_CHECK();
    const long long rows = (long long)N * D * H;
    hipLaunchKernelGGL(lncc_bwd_boxw_combine_kernel, dim3(da_grid(rows * W, 256)), dim3(256), 0, st, g2, I, J, dI, dJ, rows, W, Wo, F, dil, stride);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t da_gradloss_ws_bytes(int N, int D, int H, int W) {
    (void)D; (void)H; (void)W;
    return da_align((size_t)N * kBlocks * sizeof(double));
}

extern "C" int da_gradloss_fwd(const float* disp, int N, int D, int H, int W, const float* spacing3, int normalize, int norm,
                               float* loss, void* ws, size_t ws_bytes, void* stream) {
    if (!disp || !loss || N <= 0 || D < 3 || H < 3 || W < 3 || (norm != 1 && norm != 2)) return DA_ERR_BADARG;
    if (ws_bytes < da_gradloss_ws_bytes(N, D, H, W)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    const GradK K = gradloss_coeffs(N, D, H, W, spacing3, normalize, norm);
    const long long total = (long long)D * H * W * 3;
    int nblocks = (int)da_cdiv(total, 256 * 2); if (nblocks > kBlocks) nblocks = kBlocks; if (nblocks < 1) nblocks = 1;
    hipLaunchKernelGGL(gradloss_partial_kernel, dim3(nblocks, N), dim3(256), 0, st, disp, D, H, W, K, (double*)ws);
    DA_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nblocks * N, 1.0, 0.0, loss);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_gradloss_bwd(const float* disp, const float* dloss, float* d_disp, int N, int D, int H, int W,
                               const float* spacing3, int normalize, int norm, void* stream) {
    if (!disp || !dloss || !d_disp || N <= 0 || D < 3 || H < 3 || W < 3 || (norm != 1 && norm != 2)) return DA_ERR_BADARG;
    const GradK K = gradloss_coeffs(N, D, H, W, spacing3, normalize, norm);
    const long long total = (long long)D * H * W * 3;
    hipLaunchKernelGGL(gradloss_bwd_kernel, dim3(da_grid(total, 256, 4096), N), dim3(256), 0, da_stream(stream), disp, dloss, d_disp, D, H, W, K);
    DA_LAUNCH_CHECK();
    return 0;
}
