// Pointwise ("no spatial reuse") channel GEMMs on the matrix cores: 1x1x1 convolution and the 2x2x2 stride-2 transposed
// convolution, forward / data-gradient / weight-gradient.  Rows a3, a5 of SURVEY.md §8 (unets.py:49,55,249-250).
//
// These are HBM-bound: the largest operand is the 32-channel full-resolution tensor (629 MB per volume) and every
// element is used in one small dot product.  The VALU versions were bound by per-lane strided channel loads; here the
// A operand is read straight from global memory in MFMA fragment order (16 lanes x 16 B = one voxel's 64-byte channel
// run per instruction group), K is permuted so one 16-byte load feeds four v_mfma_f32_16x16x4_f32 (exact fp32), and the
// packed weights (<= 128 KB) stay L1/L2 resident.  Algorithmic bytes: forward = read in + write out once; dgrad = read
// dy + write dx once; wgrad = read in + dy once.
//
// map(v, tap): for the transposed conv the fine-grid voxel of coarse voxel v=(n,d,h,w) and tap (i,j,k) is
//   ((n*2D + 2d+i)*2H + 2h+j)*2W + 2w+k = fine0(v) + (i*2H + j)*2W + k.   For 1x1 convs ntaps = 1 and map = identity.
#include "common.h"
#include "pointwise_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct PwP {
    const float* a; const float* wp; const float* bias; float* out;
    long long M;            // number of COARSE voxels (rows handled by the grid)
    int D, H, W;            // coarse dims (used when up != 0)
    int K, Nc;              // GEMM inner / output channel counts of THIS launch (multiples of 16, <= 64)
    int lda, ldo;           // row strides (floats) of A and out: the full channel counts when a launch covers a 64-wide slice
    int accum;              // != 0: out += (a later K slice of the same output slice); bias is then not added again
    int ntaps, up;
    double* stats; int stats_ld;   // STATS: per-workgroup BatchNorm partial sums [gridDim.x][2][stats_ld] of this launch's Nc output columns
    const float* ps; const float* pt; float pslope;   // optional input prologue (scatter form): A is a raw tensor, act(A * ps + pt) is consumed
};

// max(z, z * s): LeakyReLU / ReLU for s in [0, 1), identity for s = 1 (same expression as the 3x3x3 kernels' prologue)
__device__ __forceinline__ float pw_act01(float z, float s) { return fmaxf(z, z * s); }

__device__ __forceinline__ long long fine0(long long v, int D, int H, int W) {
    int n, d, h, w; da_vox4(v, D, H, W, n, d, h, w);           // (32-bit divisions when the index fits: this runs 16 + 4 times per lane and chunk)
    return (((long long)n * (2 * D) + 2 * d) * (2 * H) + 2 * h) * (long long)(2 * W) + 2 * w;
}
__device__ __forceinline__ long long tapoff(int t, int H, int W) {
    return ((long long)(t >> 2) * (2 * H) + ((t >> 1) & 1)) * (2 * W) + (t & 1);
}

// SCATTER (GATHER == false): out[map(v,t)][n] = bias[n] + sum_k A[v][k] * B_t[k][n]      (conv1x1 fwd, deconv fwd)
// GATHER  (GATHER == true) : out[v][n]        = sum_t sum_k A[map(v,t)][k] * B_t[k][n]   (conv1x1 dgrad, deconv dgrad)
// STATS (SCATTER only): the epilogue also accumulates the BatchNorm sums of everything this workgroup writes (per-lane fp32 over
// the <= 16 values of a tap, then double), so the transposed conv + BatchNorm3d of unets.py:49-51 needs no statistics pass over y.
// TA / TO: storage type of the A operand / of the output (float, or da_bf16 = bf16 activation storage, common.h; the struct's pointers are
// then bf16 tensors behind a float* and lda / ldo still count ELEMENTS).  Arithmetic, bias, statistics: fp32 / double either way.
template <int KC, int NT, bool GATHER, bool STATS = false, typename TA = float, typename TO = float>
__global__ void __launch_bounds__(256) pw_mfma_kernel(PwP p) {
    static_assert(!(STATS && GATHER), "statistics are an epilogue of the scatter form");
    const TA* __restrict__ pa = reinterpret_cast<const TA*>(p.a);
    TO* __restrict__ po = reinterpret_cast<TO*>(p.out);
    constexpr int MT = 4;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const long long vbase = ((long long)blockIdx.x * 4 + wave) * (MT * 16);
    if (!STATS && vbase >= p.M) return;         // (STATS: every wave reaches the workgroup reduction; its rows are simply all invalid)
    double d1[STATS ? NT : 1], d2[STATS ? NT : 1];
#pragma unroll
    for (int n = 0; n < (STATS ? NT : 1); ++n) { d1[n] = 0.0; d2[n] = 0.0; }
    const float4* wp4 = reinterpret_cast<const float4*>(p.wp) + lane;

    long long arow[MT];            // A-row voxel (coarse; mapped to fine0 when the A side is the fine grid)
    bool aval[MT];
#pragma unroll
    for (int r = 0; r < MT; ++r) {
        const long long v = vbase + r * 16 + i;
        aval[r] = v < p.M;
        const long long vv = aval[r] ? v : 0;
        arow[r] = (GATHER && p.up) ? fine0(vv, p.D, p.H, p.W) : vv;
    }
    f32x4 acc[MT][NT];

    if constexpr (!GATHER) {
        float4 a[MT][KC];
#pragma unroll
        for (int r = 0; r < MT; ++r)
#pragma unroll
            for (int c = 0; c < KC; ++c)
                a[r][c] = aval[r] ? da_ldq(pa, (arow[r] * p.lda + 16 * c + 4 * g) >> 2) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.ps) {      // deferred BatchNorm + activation of the producer (rows past M are never stored, so they need no mask)
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const float4 sc = *reinterpret_cast<const float4*>(p.ps + 16 * c + 4 * g), sf = *reinterpret_cast<const float4*>(p.pt + 16 * c + 4 * g);
#pragma unroll
                for (int r = 0; r < MT; ++r) {
                    a[r][c].x = pw_act01(a[r][c].x * sc.x + sf.x, p.pslope); a[r][c].y = pw_act01(a[r][c].y * sc.y + sf.y, p.pslope);
                    a[r][c].z = pw_act01(a[r][c].z * sc.z + sf.z, p.pslope); a[r][c].w = pw_act01(a[r][c].w * sc.w + sf.w, p.pslope);
                    // bf16 storage: the consumer sees what a materialised (stored) activation would hold
                    if constexpr (DaEl<TA>::bf) a[r][c] = da_unpack_bf16x4(da_pack_bf16x4(a[r][c]));
                }
            }
        }
        // output rows of this lane: voxel 4*g + reg of every M-tile
        long long orow[MT][4];
#pragma unroll
        for (int r = 0; r < MT; ++r)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const long long v = vbase + r * 16 + 4 * g + reg;
                orow[r][reg] = (v < p.M) ? (p.up ? fine0(v, p.D, p.H, p.W) : v) : -1;
            }
        long long obase[MT][4];       // element offset of this lane's output column in row orow (computed once, not once per tap)
#pragma unroll
        for (int r = 0; r < MT; ++r)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) obase[r][reg] = orow[r][reg] * p.ldo + i;
        float bv[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[n] = (p.bias && !p.accum) ? p.bias[16 * n + i] : 0.f;
#pragma unroll 1
        for (int t = 0; t < p.ntaps; ++t) {
            const long long toff = p.up ? tapoff(t, p.H, p.W) : 0;
#pragma unroll
            for (int r = 0; r < MT; ++r)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[r][n] = (f32x4){bv[n], bv[n], bv[n], bv[n]};
            if (p.accum) {
#pragma unroll
                for (int r = 0; r < MT; ++r)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        if (orow[r][reg] < 0) continue;
                        const long long o = (orow[r][reg] + toff) * p.ldo + i;
#pragma unroll
                        for (int n = 0; n < NT; ++n) acc[r][n][reg] = da_ld1(po, o + 16 * n);
                    }
            }
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float4 b = wp4[(((size_t)t * NT + n) * KC + c) * 64];
#pragma unroll
                    for (int r = 0; r < MT; ++r) {
                        acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][c].x, b.x, acc[r][n], 0, 0, 0);
                        acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][c].y, b.y, acc[r][n], 0, 0, 0);
                        acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][c].z, b.z, acc[r][n], 0, 0, 0);
                        acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][c].w, b.w, acc[r][n], 0, 0, 0);
                    }
                }
            {
                TO* const pot = po + toff * p.ldo;                 // the tap's offset is wave-uniform: folded into the (scalar) base pointer
#pragma unroll
                for (int r = 0; r < MT; ++r)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        if (orow[r][reg] < 0) continue;
#pragma unroll
                        for (int n = 0; n < NT; ++n) da_st1(pot, obase[r][reg] + 16 * n, acc[r][n][reg]);
                    }
            }
            if constexpr (STATS) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    float t1 = 0.f, t2 = 0.f;
#pragma unroll
                    for (int r = 0; r < MT; ++r)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const float v = (orow[r][reg] >= 0) ? acc[r][n][reg] : 0.f;
                            t1 += v; t2 += v * v;
                        }
                    d1[n] += (double)t1; d2[n] += (double)t2;
                }
            }
        }
        if constexpr (STATS) {
            // lanes i, i+16, i+32, i+48 hold the same channel (16n + i): fold the four voxel groups, then the four waves through LDS
            __shared__ double sred[4][2][NT * 16];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                double a = d1[n], b = d2[n];
                a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
                a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
                if (g == 0) { sred[wave][0][n * 16 + i] = a; sred[wave][1][n * 16 + i] = b; }
            }
            __syncthreads();
            if ((int)threadIdx.x < NT * 16) {
                const int c = threadIdx.x;
                p.stats[((size_t)blockIdx.x * 2 + 0) * p.stats_ld + c] = (sred[0][0][c] + sred[1][0][c]) + (sred[2][0][c] + sred[3][0][c]);
                p.stats[((size_t)blockIdx.x * 2 + 1) * p.stats_ld + c] = (sred[0][1][c] + sred[1][1][c]) + (sred[2][1][c] + sred[3][1][c]);
            }
        }
    } else {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float bvn = (p.bias && !p.accum) ? p.bias[16 * n + i] : 0.f;       // only the strided conv (k2 s2) has a bias here
#pragma unroll
            for (int r = 0; r < MT; ++r) acc[r][n] = (f32x4){bvn, bvn, bvn, bvn};
        }
        if (p.accum) {
#pragma unroll
            for (int r = 0; r < MT; ++r)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const long long v = vbase + r * 16 + 4 * g + reg;
                    if (v >= p.M) continue;
                    const long long o = v * p.ldo + i;
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[r][n][reg] = da_ld1(po, o + 16 * n);
                }
        }
#pragma unroll 1
        for (int t = 0; t < p.ntaps; ++t) {
            const long long toff = p.up ? tapoff(t, p.H, p.W) : 0;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                float4 a[MT];
#pragma unroll
                for (int r = 0; r < MT; ++r)
                    a[r] = aval[r] ? da_ldq(pa, ((arow[r] + toff) * p.lda + 16 * c + 4 * g) >> 2) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float4 b = wp4[(((size_t)t * NT + n) * KC + c) * 64];
#pragma unroll
                    for (int r = 0; r < MT; ++r) {
                        acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].x, b.x, acc[r][n], 0, 0, 0);
                        acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].y, b.y, acc[r][n], 0, 0, 0);
                        acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].z, b.z, acc[r][n], 0, 0, 0);
                        acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].w, b.w, acc[r][n], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < MT; ++r)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const long long v = vbase + r * 16 + 4 * g + reg;
                if (v >= p.M) continue;
                const long long o = v * p.ldo + i;
#pragma unroll
                for (int n = 0; n < NT; ++n) da_st1(po, o + 16 * n, acc[r][n][reg]);
            }
    }
}

// packed B: wp[t][n][c][lane][m] = B_t[k = 16c + 4(lane>>4) + m][j = 16n + (lane&15)]
// transposed == 0: B_t[k][j] = w[(t*K + k)*N + j]    transposed != 0: B_t[k][j] = w[(t*N + j)*K + k]
// (K, N) is the slice [k0, k0 + K) x [n0, n0 + N) of the full Kt x Nt weight matrices
__global__ void pw_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int ntaps, int K, int N, int transposed,
                               int Kt, int Nt, int k0, int n0) {
    const int KC = K / 16, NT = N / 16;
    const long long total = (long long)ntaps * NT * KC * 256;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        long long rest = idx >> 8;
        const int c = (int)(rest % KC); rest /= KC;
        const int n = (int)(rest % NT); const int t = (int)(rest / NT);
        const int k = 16 * c + 4 * (lane >> 4) + m, j = 16 * n + (lane & 15);
        wp[idx] = transposed ? w[((size_t)t * Nt + n0 + j) * Kt + k0 + k] : w[((size_t)t * Kt + k0 + k) * Nt + n0 + j];
    }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient: dW[t][ci][co] = sum_v in[v][ci] * dy[map(v,t)][co]
// Each wave walks its own voxel range (K split over waves and blocks); blockIdx.y selects a group of TPB taps.
// ---------------------------------------------------------------------------------------------------
struct PwWgP {
    const float* in; const float* dy; float* partial;
    long long M; int D, H, W, Cin, Cout, ntaps, up;      // Cin / Cout: the (<= 64 wide) channel slices of this launch
    int ldi, ldy;                                        // row strides (floats): the full channel counts
    long long vox_per_wave;
    const float* ps; const float* pt; float pslope;      // optional input prologue on `in` (see PwP)
};

template <typename T> __device__ __forceinline__ float pw_buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    if constexpr (DaEl<T>::bf) return __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, 0) << 16);
    else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
// two consecutive bf16 channels in one 4-byte load (lo, hi)
__device__ __forceinline__ void pw_buf_ld2_bf16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float& lo, float& hi) {
    const unsigned u = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0);
    lo = __uint_as_float(u << 16); hi = __uint_as_float(u & 0xFFFF0000u);
}
// TI / TY: storage types of `in` and `dy` (see pw_mfma_kernel); ldi / ldy count elements.
// bf16 tensors with an even number of channel tiles are fetched as channel PAIRS: a 2-byte load per lane and MFMA operand made this kernel
// load-instruction-bound at half the bytes per instruction (transposed-conv weight gradient 0.76 ms against 0.36 ms on fp32 tensors).  One
// 4-byte load then feeds two operands, tile a / c of a pair holding channels 2 i + (a & 1): only the row / column -> channel map of the
// accumulators changes (chan_i / chan_y below).
template <int CIT, int COT, int TPB, typename TI = float, typename TY = float>
__global__ void __launch_bounds__(256) pw_mfma_wgrad_kernel(PwWgP p) {
    constexpr unsigned EI = DaEl<TI>::bytes, EY = DaEl<TY>::bytes;
    constexpr bool PAIRI = DaEl<TI>::bf && CIT % 2 == 0, PAIRY = DaEl<TY>::bf && COT % 2 == 0;
    auto chan_i = [](int a, int m) -> int { return PAIRI ? 32 * (a >> 1) + 2 * m + (a & 1) : 16 * a + m; };     // tile a, row m -> input channel
    auto chan_y = [](int c, int n) -> int { return PAIRY ? 32 * (c >> 1) + 2 * n + (c & 1) : 16 * c + n; };     // tile c, column n -> output channel
    extern __shared__ __attribute__((aligned(16))) float red[];      // [CIT*TPB*COT][64][4] per-block reduction
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int t0 = blockIdx.y * TPB;
    f32x4 acc[CIT][TPB][COT];
#pragma unroll
    for (int a = 0; a < CIT; ++a)
#pragma unroll
        for (int t = 0; t < TPB; ++t)
#pragma unroll
            for (int c = 0; c < COT; ++c) acc[a][t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long long toffs[TPB];
#pragma unroll
    for (int t = 0; t < TPB; ++t) toffs[t] = p.up ? tapoff(t0 + t, p.H, p.W) : 0;

    const long long vblk = (long long)blockIdx.x * 4 * p.vox_per_wave;      // first voxel of this workgroup
    const long long v0 = vblk + (long long)wave * p.vox_per_wave;
    long long v1 = v0 + p.vox_per_wave; if (v1 > p.M) v1 = p.M;
    // Branch-free operand fetch: both tensors are addressed through buffer descriptors with 32-bit byte offsets; a lane
    // past the end gets offset 0xFFFFFFFF (hardware returns 0).  All CIT + TPB*COT loads of a K-step are issued back to
    // back and the NEXT step's loads are in flight while the current step's MFMAs issue (hipcc otherwise emits
    // load -> s_waitcnt vmcnt(0) -> 4 MFMAs per tap, i.e. eight exposed memory round trips per step).
    // The descriptors are based at this WORKGROUP's first voxel (for the up-sampler's dy: at the first fine voxel of the
    // coarse d-slice that voxel lies in), so the 32-bit offsets only span one workgroup's share and tensors of any size work.
    const unsigned fine_mult = p.up ? 8u : 1u;
    const long long slice0 = p.up ? vblk / ((long long)p.H * p.W) : 0;          // n*D + d of the first voxel
    const long long fbase = p.up ? slice0 * 2 * (2ll * p.H) * (2ll * p.W) : vblk;   // first dy voxel the descriptor covers
    auto clip32 = [](unsigned long long b) -> unsigned { return b > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (unsigned)b; };
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.in) + vblk * p.ldi * (long long)EI), 0,
                                           clip32((unsigned long long)(p.M - vblk) * p.ldi * (unsigned long long)EI), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.dy) + fbase * p.ldy * (long long)EY), 0,
                                           clip32((unsigned long long)(p.M * fine_mult - fbase) * p.ldy * (unsigned long long)EY), 0x00020000);
    // this lane's voxel (v0 + g, then += 4 per K-step) tracked as (w, h, rest = n*D + d - slice0) with carries
    int cw = 0, chh = 0; long long crest = 0;
    if (p.up) {
        const long long vs = v0 + g;
        cw = (int)(vs % p.W); const long long r1 = vs / p.W;
        chh = (int)(r1 % p.H); crest = r1 / p.H - slice0;
    }
    unsigned toffb[TPB];
#pragma unroll
    for (int t = 0; t < TPB; ++t) toffb[t] = (unsigned)(toffs[t] * p.ldy * EY);
    float psc[CIT], psf[CIT];
#pragma unroll
    for (int a = 0; a < CIT; ++a) { psc[a] = p.ps ? p.ps[chan_i(a, i)] : 1.f; psf[a] = p.ps ? p.pt[chan_i(a, i)] : 0.f; }
    const bool pro = p.ps != nullptr;
    auto fetch = [&](long long vb, float* av, float (*bv)[COT], bool& okout) {
        const long long v = vb + g;
        const bool ok = v < v1;
        okout = ok;
        if constexpr (PAIRI) {
            const unsigned offa = ok ? (unsigned)(((v - vblk) * p.ldi + 2 * i) * EI) : 0xFFFFFFFFu;
#pragma unroll
            for (int a = 0; a < CIT; a += 2) pw_buf_ld2_bf16(rin, ok ? offa + 32u * EI * (a >> 1) : 0xFFFFFFFFu, av[a], av[a + 1]);
        } else {
        const unsigned offa = ok ? (unsigned)(((v - vblk) * p.ldi + i) * EI) : 0xFFFFFFFFu;
#pragma unroll
        for (int a = 0; a < CIT; ++a)
            av[a] = pw_buf_ld1<TI>(rin, ok ? offa + 16u * EI * a : 0xFFFFFFFFu);
        }
        long long fv = v - vblk;
        if (p.up) {
            fv = ((crest * 2) * (2 * p.H) + 2 * chh) * (long long)(2 * p.W) + 2 * cw;
            cw += 4;
            while (cw >= p.W) { cw -= p.W; if (++chh >= p.H) { chh = 0; ++crest; } }
        }
        if constexpr (PAIRY) {
            const unsigned offb = (unsigned)((fv * p.ldy + 2 * i) * EY);
#pragma unroll
            for (int t = 0; t < TPB; ++t)
#pragma unroll
                for (int c = 0; c < COT; c += 2)
                    pw_buf_ld2_bf16(rdy, ok ? offb + toffb[t] + 32u * EY * (c >> 1) : 0xFFFFFFFFu, bv[t][c], bv[t][c + 1]);
        } else {
        const unsigned offb = (unsigned)((fv * p.ldy + i) * EY);
#pragma unroll
        for (int t = 0; t < TPB; ++t)
#pragma unroll
            for (int c = 0; c < COT; ++c)
                bv[t][c] = pw_buf_ld1<TY>(rdy, ok ? offb + toffb[t] + 16u * EY * c : 0xFFFFFFFFu);
        }
    };
    // the deferred activation is applied when a fragment is USED (one K-step after its loads were issued), never at fetch time
    auto apply_pro = [&](float* av, bool ok) {
        if (pro) {
#pragma unroll
            for (int a = 0; a < CIT; ++a) {
                av[a] = ok ? pw_act01(av[a] * psc[a] + psf[a], p.pslope) : 0.f;
                if constexpr (DaEl<TI>::bf) av[a] = da_round_bf16(av[a]);      // (see pw_mfma_kernel)
            }
        }
    };
    float avA[CIT], bvA[TPB][COT], avB[CIT], bvB[TPB][COT];
    bool okA = false, okB = false;
    if (v0 < v1) fetch(v0, avA, bvA, okA);
#pragma unroll 1
    for (long long vb = v0; vb < v1; vb += 8) {
        fetch(vb + 4, avB, bvB, okB);               // past-the-end steps fetch zeros
        apply_pro(avA, okA);
#pragma unroll
        for (int t = 0; t < TPB; ++t)
#pragma unroll
            for (int a = 0; a < CIT; ++a)
#pragma unroll
                for (int c = 0; c < COT; ++c)
                    acc[a][t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(avA[a], bvA[t][c], acc[a][t][c], 0, 0, 0);
        fetch(vb + 8, avA, bvA, okA);
        apply_pro(avB, okB);
#pragma unroll
        for (int t = 0; t < TPB; ++t)
#pragma unroll
            for (int a = 0; a < CIT; ++a)
#pragma unroll
                for (int c = 0; c < COT; ++c)
                    acc[a][t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(avB[a], bvB[t][c], acc[a][t][c], 0, 0, 0);
    }
    // reduce the four waves through LDS, then write the block partial
    constexpr int NACC = CIT * TPB * COT;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < CIT; ++a)
#pragma unroll
                for (int t = 0; t < TPB; ++t)
#pragma unroll
                    for (int c = 0; c < COT; ++c) {
                        float4* slot = reinterpret_cast<float4*>(red) + ((a * TPB + t) * COT + c) * 64 + lane;
                        float4 cur = (w == 0) ? make_float4(0.f, 0.f, 0.f, 0.f) : *slot;
                        cur.x += acc[a][t][c][0]; cur.y += acc[a][t][c][1]; cur.z += acc[a][t][c][2]; cur.w += acc[a][t][c][3];
                        *slot = cur;
                    }
        }
        __syncthreads();
    }
    // rows (M = ci) = 4*g + reg, cols (N = co) = i
    float* part = p.partial + (size_t)blockIdx.x * p.ntaps * p.Cin * p.Cout;
    for (int idx = threadIdx.x; idx < NACC * 64; idx += 256) {
        const int ln = idx & 63; const int q = idx >> 6;
        const int c = q % COT; const int t = (q / COT) % TPB; const int a = q / (COT * TPB);
        const float4 val = reinterpret_cast<const float4*>(red)[idx];
        const int gg = ln >> 4, jj = ln & 15;
        const float vals[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ci = chan_i(a, 4 * gg + reg), co = chan_y(c, jj);
            part[((size_t)(t0 + t) * p.Cin + ci) * p.Cout + co] = vals[reg];
        }
    }
}

__global__ void pw_reduce_kernel(const float* __restrict__ partial, int nparts, int O, float* __restrict__ out) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < O; o += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int b = 0; b < nparts; ++b) s += (double)partial[(size_t)b * O + o];
        out[o] = (float)s;
    }
}

// ---------------------------------------------------------------------------------------------------
// Fused backward of an up-sampler block: BatchNorm + activation backward APPLIED ON THE FLY + transposed-conv (k2 s2) data gradient + weight gradient
// (autograd of unets.py:49-52).  Op by op the 32-channel full-resolution link moves the tensor five times beyond the reduction pass: the apply pass reads
// the incoming gradient and the raw output y and writes dy (13.5 GB per seg step in bn_act_bwd_apply alone, profiles/r06_step_traffic_seg.txt), then the
// data gradient and the weight gradient each read dy again.  Here ONE kernel reads (gout, y) once: dy = scale (dz - mean dz - xhat mean(dz xhat)),
// dz = gout act'(y scale + shift), lives in registers only,
//   dx[v][ci]     = sum_t sum_co dy[map(v, t)][co] W_t[ci][co]      M = coarse voxels, K = (tap, cout)     (GATHER form of pw_mfma_kernel)
//   dW_t[ci][co]  = sum_v in[v][ci] dy[map(v, t)][co]               K = coarse voxels                      (pw_mfma_wgrad_kernel)
//   dbias[co]     = sum dy[.][co]                                   (the transposed conv's bias: analytically zero behind a BatchNorm, computed all the same)
// A workgroup walks chunks of 32 coarse voxels; its four waves SHARE a chunk and take two taps each (the weight gradient keeps one accumulator set per
// tap: 8 taps would be 128 registers per wave).  dy is formed in the data gradient's A layout (lane = voxel, four consecutive couts), multiplied, then
// turned into the weight gradient's B layout (lane = cout, K = voxel) through a padded per-wave LDS tile (144-byte rows: conflict-free 16-byte stores and
// 4-byte column reads).  The four waves' partial dx tiles meet in LDS (two barriers per chunk), every wave owns its taps' dW outright.
// fp32 matrix instructions (exact, as the kernels it replaces); the next (chunk, tap)'s loads fly under the current one's MFMAs.
// ---------------------------------------------------------------------------------------------------
struct DbP {
    const float* gout; const float* y; const float* in; const float* wp; float* dx;
    float* dw_partial; double* col_partial;
    const float* mean; const float* rstd; const float* scale; const float* shift; const float* cm; float slope;
    long long M, nchunks; int D, H, W;
};

template <int CTRL> __device__ __forceinline__ double pw_dpp_add_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

template <int KC, int NT>
__global__ void __launch_bounds__(256, 2) deconv_bn_bwd_kernel(DbP p) {
    constexpr int MT = 2;                                   // M-tiles (16 coarse voxels) per chunk
    static_assert(MT * NT == 4, "one dx tile per wave in the cross-wave reduction");
    constexpr int COUT = 16 * KC, CIN = 16 * NT, TRS = COUT + 4;
    __shared__ __attribute__((aligned(16))) float tr[4][MT * 16 * TRS];
    __shared__ __attribute__((aligned(16))) float red2[2][4][MT * NT][64 * 4];      // two copies, alternating by chunk: ONE barrier per chunk (a copy is rewritten two barriers after it was read)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4;
    const int t0 = 2 * wave;
    int par = 0;
    float k_sc[KC][4], k_sf[KC][4], k_mu[KC][4], k_c1[KC][4], k_c2[KC][4];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ch = 16 * c + 4 * g + m;
            k_sc[c][m] = p.scale[ch]; k_sf[c][m] = p.shift[ch]; k_mu[c][m] = p.mean[ch];
            k_c1[c][m] = p.scale[ch] * p.cm[ch]; k_c2[c][m] = p.scale[ch] * p.rstd[ch] * p.cm[COUT + ch];
        }
    f32x4 acc_dw[2][NT][KC];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int c = 0; c < KC; ++c) acc_dw[tt][a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float colacc[KC][4];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) colacc[c][m] = 0.f;
    const float4* wp4 = reinterpret_cast<const float4*>(p.wp) + lane;
    const long long toffA = tapoff(t0, p.H, p.W), toffB = tapoff(t0 + 1, p.H, p.W);
    float* trw = tr[wave];

    // unconditional loads (a row past the end reads coarse voxel 0 and is masked when dy is formed): nothing that touches memory sits in a branch
    auto rows = [&](long long chunk, long long (&frow)[MT], bool (&aval)[MT]) {
#pragma unroll
        for (int r = 0; r < MT; ++r) {
            const long long v = chunk * (MT * 16) + r * 16 + i;
            aval[r] = v < p.M;
            frow[r] = fine0(aval[r] ? v : 0, p.D, p.H, p.W);
        }
    };
    auto load_raw = [&](const long long (&frow)[MT], long long tf, float4 (&gq)[MT][KC], float4 (&yq)[MT][KC]) {
#pragma unroll
        for (int r = 0; r < MT; ++r)
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const long long q = ((frow[r] + tf) * COUT + 16 * c + 4 * g) >> 2;
#if defined(DA_DB_ABL) && (DA_DB_ABL & 4)
                gq[r][c] = make_float4((float)q, 1.f, 2.f, 3.f); yq[r][c] = make_float4(1.f, (float)q, 2.f, 3.f);      // timing only
#else
                gq[r][c] = da_ldq_nt(p.gout, q); yq[r][c] = da_ldq_nt(p.y, q);
#endif
            }
    };
    auto load_in = [&](long long chunk, float (&ain)[MT][4][NT]) {     // A operand of the weight gradient: lane (i, g) holds in[voxel 16 r + 4 g + m][16 a + i]
#pragma unroll
        for (int r = 0; r < MT; ++r)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const long long v = chunk * (MT * 16) + r * 16 + 4 * g + m;
                const bool ok = v < p.M;
                const long long vv = ok ? v : 0;
#pragma unroll
                for (int a = 0; a < NT; ++a) { const float t = p.in[vv * CIN + 16 * a + i]; ain[r][m][a] = ok ? t : 0.f; }
            }
    };
    auto compute = [&](int tt, const float4 (&gq)[MT][KC], const float4 (&yq)[MT][KC], const bool (&aval)[MT], const float (&ain)[MT][4][NT], f32x4 (&acc_dx)[MT][NT]) {
        float4 dq[MT][KC];
#pragma unroll
        for (int r = 0; r < MT; ++r)
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const float gv[4] = {gq[r][c].x, gq[r][c].y, gq[r][c].z, gq[r][c].w}, xv[4] = {yq[r][c].x, yq[r][c].y, yq[r][c].z, yq[r][c].w};
                float o[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float z = xv[m] * k_sc[c][m] + k_sf[c][m];
                    const float dz = gv[m] * da_act_grad(z, p.slope);
                    const float t = k_sc[c][m] * dz - k_c1[c][m] - (xv[m] - k_mu[c][m]) * k_c2[c][m];      // bn_act_bwd_apply_kernel's expression (norm_act.hip)
                    o[m] = aval[r] ? t : 0.f;
                    colacc[c][m] += o[m];
                }
                dq[r][c] = make_float4(o[0], o[1], o[2], o[3]);
            }
        // data gradient: A = dy (row = coarse voxel i, K = cout 16 c + 4 g + m), B = packed W_t (K = cout, column = cin 16 n + i)
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float4 b = wp4[(((size_t)(t0 + tt) * NT + n) * KC + c) * 64];
#pragma unroll
                for (int r = 0; r < MT; ++r) {
#if defined(DA_DB_ABL) && (DA_DB_ABL & 1)
                    acc_dx[r][n][0] += dq[r][c].x * b.x + dq[r][c].y * b.y + dq[r][c].z * b.z + dq[r][c].w * b.w;      // timing only
#else
                    acc_dx[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(dq[r][c].x, b.x, acc_dx[r][n], 0, 0, 0);
                    acc_dx[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(dq[r][c].y, b.y, acc_dx[r][n], 0, 0, 0);
                    acc_dx[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(dq[r][c].z, b.z, acc_dx[r][n], 0, 0, 0);
                    acc_dx[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(dq[r][c].w, b.w, acc_dx[r][n], 0, 0, 0);
#endif
                }
            }
#if defined(DA_DB_ABL) && (DA_DB_ABL & 2)
        if (p.slope == 12345.f)      // timing only: no weight-gradient half
#endif
        // dy into the weight gradient's B layout through this wave's LDS tile
#pragma unroll
        for (int r = 0; r < MT; ++r)
#pragma unroll
            for (int c = 0; c < KC; ++c) *reinterpret_cast<float4*>(trw + (16 * r + i) * TRS + 16 * c + 4 * g) = dq[r][c];
        // weight gradient: A = in (row = cin 16 a + i, K = voxel 16 r + 4 g + m), B = dy (K = voxel, column = cout 16 c + i)
#if defined(DA_DB_ABL) && (DA_DB_ABL & 2)
        if (p.slope == 12345.f)
#endif
#pragma unroll
        for (int r = 0; r < MT; ++r)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float bd[KC];
#pragma unroll
                for (int c = 0; c < KC; ++c) bd[c] = trw[(16 * r + 4 * g + m) * TRS + 16 * c + i];
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int c = 0; c < KC; ++c)
#if defined(DA_DB_ABL) && (DA_DB_ABL & 1)
                        acc_dw[tt][a][c][0] += ain[r][m][a] * bd[c];      // timing only
#else
                        acc_dw[tt][a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ain[r][m][a], bd[c], acc_dw[tt][a][c], 0, 0, 0);
#endif
            }
    };

    long long chunk = blockIdx.x;
    long long frow[MT]; bool aval[MT];
    float4 g0[MT][KC], y0[MT][KC], g1[MT][KC], y1[MT][KC];
    float ain[MT][4][NT];
    if (chunk < p.nchunks) { rows(chunk, frow, aval); load_raw(frow, toffA, g0, y0); load_in(chunk, ain); }
#pragma unroll 1
    while (chunk < p.nchunks) {
        f32x4 acc_dx[MT][NT];
#pragma unroll
        for (int r = 0; r < MT; ++r)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc_dx[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        load_raw(frow, toffB, g1, y1);                        // this chunk's second tap
        compute(0, g0, y0, aval, ain, acc_dx);
        const long long next = chunk + gridDim.x;
        const bool has = next < p.nchunks;
        long long frow2[MT]; bool aval2[MT];
        rows(has ? next : chunk, frow2, aval2);
        load_raw(frow2, toffA, g0, y0);                       // the next chunk's first tap (the last iteration re-reads its own chunk: harmless)
        float ain2[MT][4][NT];
        load_in(has ? next : chunk, ain2);
        compute(1, g1, y1, aval, ain, acc_dx);
        // the four waves' partial dx tiles (two taps each) -> LDS -> wave q sums tile q and stores it
#if defined(DA_DB_ABL) && (DA_DB_ABL & 8)
        if (acc_dx[0][0][0] == 12345.678f)      // timing only: no cross-wave sum, no dx store, no barriers
        {
#endif
#pragma unroll
        for (int r = 0; r < MT; ++r)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                *reinterpret_cast<float4*>(&red2[par][wave][r * NT + n][lane * 4]) = make_float4(acc_dx[r][n][0], acc_dx[r][n][1], acc_dx[r][n][2], acc_dx[r][n][3]);
        __syncthreads();
        {
            const int r = wave / NT, n = wave % NT;
            float4 s = *reinterpret_cast<const float4*>(&red2[par][0][wave][lane * 4]);
#pragma unroll
            for (int w = 1; w < 4; ++w) { const float4 t = *reinterpret_cast<const float4*>(&red2[par][w][wave][lane * 4]); s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
            const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const long long v = chunk * (MT * 16) + r * 16 + 4 * g + reg;
                if (v < p.M) p.dx[v * CIN + 16 * n + i] = sv[reg];
            }
        }
        par ^= 1;
#if defined(DA_DB_ABL) && (DA_DB_ABL & 8)
        }
#endif
        chunk = next;
#pragma unroll
        for (int r = 0; r < MT; ++r) {
            frow[r] = frow2[r]; aval[r] = aval2[r];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int a = 0; a < NT; ++a) ain[r][m][a] = ain2[r][m][a];
        }
    }
    // this workgroup's dW partial: every wave writes its two taps (rows = cin 4 g + reg, columns = cout i)
    float* part = p.dw_partial + (size_t)blockIdx.x * 8 * CIN * COUT;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    part[((size_t)(t0 + tt) * CIN + 16 * a + 4 * g + reg) * COUT + 16 * c + i] = acc_dw[tt][a][c][reg];
    // column sums of dy: over the 16 voxel lanes of a row (DPP, double), then the four waves through LDS
    __syncthreads();                                               // (the last chunk's tiles are still being read)
    double* cred = reinterpret_cast<double*>(&red2[0][0][0][0]);      // [4 waves][COUT]
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            double v = (double)colacc[c][m];
            v = pw_dpp_add_f64<0xB1>(v); v = pw_dpp_add_f64<0x4E>(v); v = pw_dpp_add_f64<0x141>(v); v = pw_dpp_add_f64<0x140>(v);
            if (i == 0) cred[wave * COUT + 16 * c + 4 * g + m] = v;
        }
    __syncthreads();
    if ((int)threadIdx.x < COUT) p.col_partial[((size_t)blockIdx.x * 2) * COUT + threadIdx.x] = (cred[threadIdx.x] + cred[COUT + threadIdx.x]) + (cred[2 * COUT + threadIdx.x] + cred[3 * COUT + threadIdx.x]);
}

template <int KC, int NT, typename TA, typename TO>
int launch_pw_t(const PwP& p, bool gather, hipStream_t st) {
    const unsigned grid = (unsigned)da_cdiv(p.M, 256);
    if (gather) hipLaunchKernelGGL((pw_mfma_kernel<KC, NT, true, false, TA, TO>), dim3(grid), dim3(256), 0, st, p);
    else if (p.stats) hipLaunchKernelGGL((pw_mfma_kernel<KC, NT, false, true, TA, TO>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((pw_mfma_kernel<KC, NT, false, false, TA, TO>), dim3(grid), dim3(256), 0, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}
template <int KC, int NT>
int launch_pw(const PwP& p, bool gather, hipStream_t st, int a_bf, int o_bf) {
    if (a_bf && o_bf) return launch_pw_t<KC, NT, da_bf16, da_bf16>(p, gather, st);
    if (a_bf) return launch_pw_t<KC, NT, da_bf16, float>(p, gather, st);
    if (o_bf) return launch_pw_t<KC, NT, float, da_bf16>(p, gather, st);
    return launch_pw_t<KC, NT, float, float>(p, gather, st);
}

}  // namespace

bool da_pw_supported(int K, int N) {
    return K % 16 == 0 && N % 16 == 0 && K >= 16 && K <= 1024 && N >= 16 && N <= 1024;
}

size_t da_pw_packed_bytes(int ntaps, int K, int N) { return da_align((size_t)ntaps * (K < 64 ? K : 64) * (N < 64 ? N : 64) * sizeof(float)); }

// One launch covers a K x N slice of at most 64 x 64 channels with everything in registers; wider layers (the full UNet's
// 128 - 512 channel up-samplers, unets.py:88,91,94) are tiled on the host: for every 64-wide output slice the K slices are
// accumulated into it in stream order (accum: the accumulators start from the current output instead of the bias).
int da_pw_gemm(const float* a, const float* w, int transposed, const float* bias, float* out,
               long long M, int D, int H, int W, int K, int N, int ntaps, int up, int gather,
               void* ws, size_t ws_bytes, hipStream_t st, double* stats_partial, const float* pro_scale, const float* pro_shift, float pro_slope, int a_bf, int o_bf) {
    if (!da_pw_supported(K, N)) return DA_ERR_UNSUPPORTED;
    if ((stats_partial || pro_scale) && gather) return DA_ERR_BADARG;
    if (pro_scale && (!pro_shift || pro_slope >= 1.f)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_pw_packed_bytes(ntaps, K, N)) return DA_ERR_WS_SMALL;
    float* wp = (float*)ws;
    for (int n0 = 0; n0 < N; n0 += 64) {
        const int nn = (N - n0) < 64 ? (N - n0) : 64;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int kk = (K - k0) < 64 ? (K - k0) : 64;
            const long long total = (long long)ntaps * kk * nn;
            hipLaunchKernelGGL(pw_pack_kernel, dim3(da_grid(total, 256, 512)), dim3(256), 0, st, w, wp, ntaps, kk, nn, transposed, K, N, k0, n0);
            DA_LAUNCH_CHECK();
            PwP p;
            p.a = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a) + (size_t)k0 * (a_bf ? 2 : 4)); p.wp = wp; p.bias = bias ? bias + n0 : nullptr;
            p.out = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + (size_t)n0 * (o_bf ? 2 : 4)); p.M = M; p.D = D; p.H = H; p.W = W;
            p.K = kk; p.Nc = nn; p.lda = K; p.ldo = N; p.accum = k0 > 0 ? 1 : 0; p.ntaps = ntaps; p.up = up;
            p.stats = (stats_partial && k0 + 64 >= K) ? stats_partial + n0 : nullptr; p.stats_ld = N;      // statistics of the FINAL values: last K slice
            p.ps = pro_scale ? pro_scale + k0 : nullptr; p.pt = pro_scale ? pro_shift + k0 : nullptr; p.pslope = pro_slope < 0.f ? 1.f : pro_slope;
            const int KC = kk / 16, NT = nn / 16;
            int rc = DA_ERR_UNSUPPORTED;
#define DA_PW_CASE(kc, nt) if (KC == kc && NT == nt) rc = launch_pw<kc, nt>(p, gather != 0, st, a_bf, o_bf)
            DA_PW_CASE(1, 1); DA_PW_CASE(1, 2); DA_PW_CASE(2, 1); DA_PW_CASE(2, 2); DA_PW_CASE(4, 4);
            DA_PW_CASE(1, 4); DA_PW_CASE(4, 1); DA_PW_CASE(2, 4); DA_PW_CASE(4, 2);
            DA_PW_CASE(3, 3); DA_PW_CASE(1, 3); DA_PW_CASE(3, 1); DA_PW_CASE(2, 3); DA_PW_CASE(3, 2); DA_PW_CASE(3, 4); DA_PW_CASE(4, 3);
#undef DA_PW_CASE
            if (rc) return rc;
        }
    }
    return 0;
}

static int wg_blocks(long long M, size_t O, long long* vox_per_wave) {
    long long blocks = 1024;                      // 4 workgroups per CU (accumulators <= 128 VGPRs)
    const long long cap = (long long)((64ull << 20) / (O * 4)); if (blocks > cap) blocks = cap < 1 ? 1 : cap;
    long long vpw = da_cdiv(M, blocks * 4);
    vpw = da_cdiv(vpw, 4) * 4;
    if (vpw < 4) vpw = 4;
    *vox_per_wave = vpw;
    return (int)da_cdiv(M, vpw * 4);
}

size_t da_pw_wgrad_ws_bytes(long long M, int ntaps, int Cin, int Cout) {
    long long vpw;
    const size_t O = (size_t)ntaps * (Cin < 64 ? Cin : 64) * (Cout < 64 ? Cout : 64);      // one 64 x 64 slice at a time
    const int nb = wg_blocks(M, O, &vpw);
    return da_align((size_t)nb * O * sizeof(float)) + da_align(O * sizeof(float));
}

__global__ void pw_place_kernel(const float* __restrict__ src, float* __restrict__ dw, int ntaps, int cic, int coc, int Cin, int Cout, int ci0, int co0) {
    const int total = ntaps * cic * coc;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % coc; const int ci = (i / coc) % cic; const int t = i / (coc * cic);
        dw[((size_t)t * Cin + ci0 + ci) * Cout + co0 + co] = src[i];
    }
}

static int g_wg_in_bf = 0, g_wg_dy_bf = 0;      // storage types of the weight-gradient call being dispatched (set by da_pw_wgrad; one host thread drives the launches)
template <int CIT, int COT, int TPB, typename TI, typename TY>
static int launch_pw_wgrad_t(const PwWgP& p, int nblocks, hipStream_t st) {
    const size_t shm = (size_t)CIT * TPB * COT * 64 * 4 * sizeof(float);
    auto kern = pw_mfma_wgrad_kernel<CIT, COT, TPB, TI, TY>;
    static bool attr_set = false;
    if (!attr_set && shm > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblocks, p.ntaps / TPB), dim3(256), shm, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}
template <int CIT, int COT, int TPB>
static int launch_pw_wgrad(const PwWgP& p, int nblocks, hipStream_t st) {
    if (g_wg_in_bf && g_wg_dy_bf) return launch_pw_wgrad_t<CIT, COT, TPB, da_bf16, da_bf16>(p, nblocks, st);
    if (g_wg_in_bf) return launch_pw_wgrad_t<CIT, COT, TPB, da_bf16, float>(p, nblocks, st);
    if (g_wg_dy_bf) return DA_ERR_UNSUPPORTED;          // (fp32 input with a bf16 gradient does not occur)
    return launch_pw_wgrad_t<CIT, COT, TPB, float, float>(p, nblocks, st);
}

struct PwPro { const float* s; const float* t; float slope; };
static int pw_wgrad_slice(const float* in, const float* dy, float* dw, long long M, int D, int H, int W, int Cin, int Cout, int ldi, int ldy,
                          int ntaps, int up, void* ws, hipStream_t st, const PwPro& pro, int ci0);

int da_pw_wgrad(const float* in, const float* dy, float* dw, long long M, int D, int H, int W, int Cin, int Cout,
                int ntaps, int up, void* ws, size_t ws_bytes, hipStream_t st, const float* pro_scale, const float* pro_shift, float pro_slope, int in_bf, int dy_bf) {
    if (!da_pw_supported(Cin, Cout) || (ntaps != 1 && ntaps != 8)) return DA_ERR_UNSUPPORTED;
    g_wg_in_bf = in_bf; g_wg_dy_bf = dy_bf;
    if (pro_scale && (!pro_shift || pro_slope >= 1.f)) return DA_ERR_UNSUPPORTED;
    const PwPro pro = {pro_scale, pro_shift, pro_slope < 0.f ? 1.f : pro_slope};
    if (ws_bytes < da_pw_wgrad_ws_bytes(M, ntaps, Cin, Cout)) return DA_ERR_WS_SMALL;
    if (Cin <= 64 && Cout <= 64) return pw_wgrad_slice(in, dy, dw, M, D, H, W, Cin, Cout, Cin, Cout, ntaps, up, ws, st, pro, 0);
    // wide layers: independent 64 x 64 channel slices, each reduced into a dense scratch and placed into dW
    long long vpw;
    const size_t Os = (size_t)ntaps * 64 * 64;
    const int nbmax = wg_blocks(M, Os, &vpw);
    float* tmp = (float*)((char*)ws + da_align((size_t)nbmax * Os * sizeof(float)));
    for (int ci0 = 0; ci0 < Cin; ci0 += 64)
        for (int co0 = 0; co0 < Cout; co0 += 64) {
            const int cic = (Cin - ci0) < 64 ? (Cin - ci0) : 64, coc = (Cout - co0) < 64 ? (Cout - co0) : 64;
            const int rc = pw_wgrad_slice(reinterpret_cast<const float*>(reinterpret_cast<const char*>(in) + (size_t)ci0 * (in_bf ? 2 : 4)),
                                          reinterpret_cast<const float*>(reinterpret_cast<const char*>(dy) + (size_t)co0 * (dy_bf ? 2 : 4)),
                                          tmp, M, D, H, W, cic, coc, Cin, Cout, ntaps, up, ws, st, pro, ci0);
            if (rc) return rc;
            hipLaunchKernelGGL(pw_place_kernel, dim3(da_grid((long long)ntaps * cic * coc, 256, 256)), dim3(256), 0, st, tmp, dw, ntaps, cic, coc, Cin, Cout, ci0, co0);
            DA_LAUNCH_CHECK();
        }
    return 0;
}

static int pw_wgrad_slice(const float* in, const float* dy, float* dw, long long M, int D, int H, int W, int Cin, int Cout, int ldi, int ldy,
                          int ntaps, int up, void* ws, hipStream_t st, const PwPro& pro, int ci0) {
    PwWgP p;
    p.ps = pro.s ? pro.s + ci0 : nullptr; p.pt = pro.s ? pro.t + ci0 : nullptr; p.pslope = pro.slope;
    p.in = in; p.dy = dy; p.partial = (float*)ws; p.M = M; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.ntaps = ntaps; p.up = up;
    p.ldi = ldi; p.ldy = ldy;
    const size_t O = (size_t)ntaps * Cin * Cout;
    const int nb = wg_blocks(M, O, &p.vox_per_wave);
    const int CIT = Cin / 16, COT = Cout / 16;
    int rc = DA_ERR_UNSUPPORTED;
    // taps per block chosen so that CIT*COT*TPB*4 accumulator registers <= 128
#define DA_WG_CASE(a, c, t) if (CIT == a && COT == c && ((ntaps == 8 && t > 0) || (ntaps == 1 && t == 0))) rc = launch_pw_wgrad<a, c, (t > 0 ? t : 1)>(p, nb, st)
    if (ntaps == 8) {
        if (CIT * COT <= 4) {
            if (CIT == 1 && COT == 1) rc = launch_pw_wgrad<1, 1, 8>(p, nb, st);
            else if (CIT == 1 && COT == 2) rc = launch_pw_wgrad<1, 2, 8>(p, nb, st);
            else if (CIT == 2 && COT == 1) rc = launch_pw_wgrad<2, 1, 8>(p, nb, st);
            else if (CIT == 2 && COT == 2) rc = launch_pw_wgrad<2, 2, 8>(p, nb, st);
            else if (CIT == 1 && COT == 4) rc = launch_pw_wgrad<1, 4, 8>(p, nb, st);
            else if (CIT == 4 && COT == 1) rc = launch_pw_wgrad<4, 1, 8>(p, nb, st);
            else if (CIT == 1 && COT == 3) rc = launch_pw_wgrad<1, 3, 8>(p, nb, st);
            else if (CIT == 3 && COT == 1) rc = launch_pw_wgrad<3, 1, 8>(p, nb, st);
        } else if (CIT * COT <= 8) {
            if (CIT == 2 && COT == 4) rc = launch_pw_wgrad<2, 4, 4>(p, nb, st);
            else if (CIT == 4 && COT == 2) rc = launch_pw_wgrad<4, 2, 4>(p, nb, st);
            else if (CIT == 2 && COT == 3) rc = launch_pw_wgrad<2, 3, 4>(p, nb, st);
            else if (CIT == 3 && COT == 2) rc = launch_pw_wgrad<3, 2, 4>(p, nb, st);
        } else {
            if (CIT == 4 && COT == 4) rc = launch_pw_wgrad<4, 4, 2>(p, nb, st);
            else if (CIT == 3 && COT == 3) rc = launch_pw_wgrad<3, 3, 2>(p, nb, st);
            else if (CIT == 3 && COT == 4) rc = launch_pw_wgrad<3, 4, 2>(p, nb, st);
            else if (CIT == 4 && COT == 3) rc = launch_pw_wgrad<4, 3, 2>(p, nb, st);
        }
    } else {
        if (CIT == 1 && COT == 1) rc = launch_pw_wgrad<1, 1, 1>(p, nb, st);
        else if (CIT == 1 && COT == 2) rc = launch_pw_wgrad<1, 2, 1>(p, nb, st);
        else if (CIT == 2 && COT == 1) rc = launch_pw_wgrad<2, 1, 1>(p, nb, st);
        else if (CIT == 2 && COT == 2) rc = launch_pw_wgrad<2, 2, 1>(p, nb, st);
        else if (CIT == 4 && COT == 4) rc = launch_pw_wgrad<4, 4, 1>(p, nb, st);
        else if (CIT == 2 && COT == 4) rc = launch_pw_wgrad<2, 4, 1>(p, nb, st);
        else if (CIT == 4 && COT == 2) rc = launch_pw_wgrad<4, 2, 1>(p, nb, st);
        else if (CIT == 1 && COT == 4) rc = launch_pw_wgrad<1, 4, 1>(p, nb, st);
        else if (CIT == 4 && COT == 1) rc = launch_pw_wgrad<4, 1, 1>(p, nb, st);
    }
#undef DA_WG_CASE
    if (rc) return rc;
    { const int rc2 = da_reduce_partials(p.partial, nb, (int)O, dw, st); if (rc2) return rc2; }
    return 0;
}


// ---- fused BatchNorm-backward + transposed-conv backward (deconv_bn_bwd_kernel) ----
bool da_deconv_bn_bwd_supported(int Cin, int Cout) { return Cin == 32 && Cout == 32; }      // one (cout, cin) tile pair per wave pair: the full-resolution up-sampler of UNet_light
static int db_blocks(long long nchunks) { return (int)(nchunks < 512 ? nchunks : 512); }    // two workgroups per CU
size_t da_deconv_bn_bwd_ws_bytes(long long M, int Cin, int Cout) {
    const long long nchunks = da_cdiv(M, 32);
    const int nb = db_blocks(nchunks);
    return da_pw_packed_bytes(8, Cout, Cin) + da_align((size_t)nb * 8 * Cin * Cout * sizeof(float)) + da_align((size_t)nb * 2 * Cout * sizeof(double));
}
int da_deconv_bn_bwd(const float* gout, const float* y, const float* mean, const float* rstd, const float* scale, const float* shift, const float* cm, float slope,
                     const float* in, const float* w_tio, float* dx, float* dw_tio, float* dbias,
                     long long M, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!da_deconv_bn_bwd_supported(Cin, Cout)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_deconv_bn_bwd_ws_bytes(M, Cin, Cout)) return DA_ERR_WS_SMALL;
    if ((unsigned long long)M * 8ull * (unsigned long long)Cout >= (1ull << 62)) return DA_ERR_UNSUPPORTED;
    const long long nchunks = da_cdiv(M, 32);
    const int nb = db_blocks(nchunks);
    float* wp = (float*)ws;
    float* dwp = (float*)((char*)ws + da_pw_packed_bytes(8, Cout, Cin));
    double* colp = (double*)((char*)dwp + da_align((size_t)nb * 8 * Cin * Cout * sizeof(float)));
    // B_t[k = cout][j = cin] = w_tio[(t * Cin + cin) * Cout + cout]: the data gradient's packing (da_deconv_k2s2_dgrad)
    hipLaunchKernelGGL(pw_pack_kernel, dim3(da_grid((long long)8 * Cin * Cout, 256, 512)), dim3(256), 0, st, w_tio, wp, 8, Cout, Cin, 1, Cout, Cin, 0, 0);
    DA_LAUNCH_CHECK();
    DbP p;
    p.gout = gout; p.y = y; p.in = in; p.wp = wp; p.dx = dx; p.dw_partial = dwp; p.col_partial = colp;
    p.mean = mean; p.rstd = rstd; p.scale = scale; p.shift = shift; p.cm = cm; p.slope = slope;
    p.M = M; p.nchunks = nchunks; p.D = D; p.H = H; p.W = W;
    hipLaunchKernelGGL((deconv_bn_bwd_kernel<2, 2>), dim3(nb), dim3(256), 0, st, p);
    DA_LAUNCH_CHECK();
    { const int rc = da_reduce_partials(dwp, nb, 8 * Cin * Cout, dw_tio, st); if (rc) return rc; }
    if (dbias) return da_colsum_finish((const void*)colp, nb, Cout, dbias, 0, (void*)st);
    return 0;
}
