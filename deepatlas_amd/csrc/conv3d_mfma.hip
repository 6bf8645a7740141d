// Implicit-GEMM 3x3x3 convolution on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32).
//
// Forward / data-gradient  (da_conv3_mfma_fwd):
//   GEMM view  M = output voxels, N = Cout, K = 27 taps x Cin.   One workgroup (4 waves) owns a 4x8x16 output tile;
//   the input tile + halo (6x10x18 voxels x CK channels, NDHWC so a voxel's channels are one 16/32/64-byte run) is
//   staged in LDS once per channel chunk and re-read 27 times with shifted windows (ds_read_b128).  Wave w owns
//   z-slab w: 8 M-tiles (rows of 16 voxels along W) x NREP N-tiles of 16 couts -> 8*NREP*4 accumulator VGPRs.
//   K ordering is permuted so that one 16-byte LDS read feeds four consecutive MFMAs: in MFMA m of a K-step, lane
//   group g (= lane>>4) supplies cin = 4*g + m.  Weights are pre-packed to match ([chunk][step][ntile][lane][m]) so
//   the B fragment is one coalesced 1 KiB global (L1/L2-resident) load per wave per step, prefetched a step ahead.
//   The data gradient is the same kernel with tap-flipped / channel-transposed packing (stride 1).
//   Concat inputs are never materialised: a channel chunk reads from in1 or in2; split outputs go to out1/out2.
//
// Weight gradient (da_conv3_mfma_wgrad):
//   GEMM view  M = Cin chunk (16), N = Cout tile (16), K = voxels.  Persistent workgroups walk a slab of 2x8x16
//   tiles keeping dW partials in accumulators (wave w owns taps w, w+4, ...), stage the input halo tile and the dY
//   tile in LDS, and write one partial dW per slab; a second launch reduces the slabs in double (deterministic).
//
// Roofline: both are MFMA-bound (fp32 matrix peak 157.3 TFLOP/s); algorithmic FLOPs = 2*27*Cin*Cout per output voxel.
#include "common.h"
#include "conv3d_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TY = 8, TX = 16, HY = TY + 2, HX = TX + 2;

// bijective XCD-aware remap: consecutive tiles land on the same XCD (shared halos hit that XCD's L2)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, loc = bid / 8;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

// stage an (HZ x HY x HX) halo tile of CK channels into LDS as [voxel][CK]; out-of-volume voxels are zero (padding 1)
template <int CK, int HZ>
__device__ __forceinline__ void stage_tile(float* __restrict__ lds, const float* __restrict__ src, int Cs, int choff,
                                           int n, int z0, int y0, int x0, int D, int H, int W) {
    constexpr int Q = CK / 4;
    constexpr int TOTAL = HZ * HY * HX * Q;
#pragma unroll 4
    for (int idx = threadIdx.x; idx < TOTAL; idx += 256) {
        const int c4 = idx % Q; const int hv = idx / Q;
        const int hx = hv % HX; const int t = hv / HX;
        const int hy = t % HY; const int hz = t / HY;
        const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
            v = *reinterpret_cast<const float4*>(src + ((((long long)n * D + z) * H + y) * W + x) * Cs + choff + c4 * 4);
        *reinterpret_cast<float4*>(lds + hv * CK + c4 * 4) = v;
    }
}

struct FwdP {
    const float* in1; const float* in2; int C1, C2;
    const float* wp; const float* bias;
    float* out1; float* out2; int Cs1, Cs2;
    int N, D, H, W, Cout, NT, ntz, nty, ntx, ntiles;
    float slope;
};

template <int CK, int NREP>
__global__ void __launch_bounds__(256, 2) conv3_mfma_fwd_kernel(FwdP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TZ = 4, HZ = TZ + 2;
    constexpr int NSTEPS = (27 * CK + 15) / 16;          // 27 (CK = 16) | 14 (CK = 8: two taps per K-step)
    int t = xcd_remap(blockIdx.x, p.ntiles);
    const int tx = t % p.ntx; t /= p.ntx;
    const int ty = t % p.nty; t /= p.nty;
    const int tz = t % p.ntz; const int n = t / p.ntz;
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4;
    const int nt0 = blockIdx.y * NREP;

    f32x4 acc[TY][NREP];
#pragma unroll
    for (int r = 0; r < TY; ++r)
#pragma unroll
        for (int nn = 0; nn < NREP; ++nn) acc[r][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = (p.C1 + p.C2) / CK;
    const float4* wp4 = reinterpret_cast<const float4*>(p.wp);

    for (int ch = 0; ch < nchunks; ++ch) {
        const int cbase = ch * CK;
        const float* src; int Cs, choff;
        if (cbase < p.C1) { src = p.in1; Cs = p.C1; choff = cbase; } else { src = p.in2; Cs = p.C2; choff = cbase - p.C1; }
        __syncthreads();
        stage_tile<CK, HZ>(lds, src, Cs, choff, n, z0, y0, x0, p.D, p.H, p.W);
        __syncthreads();
        const float4* wch = wp4 + ((size_t)ch * NSTEPS * p.NT + nt0) * 64 + lane;
        float4 bcur[NREP];
#pragma unroll
        for (int nn = 0; nn < NREP; ++nn) bcur[nn] = wch[(size_t)nn * 64];

        if constexpr (CK == 16) {
            const float* abase = lds + ((wave * HY) * HX + i) * CK + g * 4;
#pragma unroll 1
            for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
                for (int dyx = 0; dyx < 9; ++dyx) {
                    const int dy = dyx / 3, dx = dyx % 3;
                    const int s = dz * 9 + dyx;
                    const int sn = (s + 1 < 27) ? s + 1 : 26;
                    float4 bnext[NREP];
#pragma unroll
                    for (int nn = 0; nn < NREP; ++nn) bnext[nn] = wch[((size_t)sn * p.NT + nn) * 64];
                    const float* ap = abase + ((dz * HY + dy) * HX + dx) * CK;
#pragma unroll
                    for (int r = 0; r < TY; ++r) {
                        const float4 a = *reinterpret_cast<const float4*>(ap + r * (HX * CK));
#pragma unroll
                        for (int nn = 0; nn < NREP; ++nn) {
                            acc[r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bcur[nn].x, acc[r][nn], 0, 0, 0);
                            acc[r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bcur[nn].y, acc[r][nn], 0, 0, 0);
                            acc[r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bcur[nn].z, acc[r][nn], 0, 0, 0);
                            acc[r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bcur[nn].w, acc[r][nn], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int nn = 0; nn < NREP; ++nn) bcur[nn] = bnext[nn];
                }
            }
        } else {
            // CK == 8: lane groups 0,1 take tap 2s (cin 0-3 / 4-7), groups 2,3 take tap 2s+1
            const float* abase = lds + ((wave * HY) * HX + i) * CK + (g & 1) * 4;
            const bool hi = (g >> 1) != 0;
#pragma unroll
            for (int s = 0; s < NSTEPS; ++s) {
                const int sn = (s + 1 < NSTEPS) ? s + 1 : NSTEPS - 1;
                float4 bnext[NREP];
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn) bnext[nn] = wch[((size_t)sn * p.NT + nn) * 64];
                const int t0 = 2 * s, t1 = (2 * s + 1 < 27) ? 2 * s + 1 : 26;
                const int off0 = (((t0 / 9) * HY + (t0 / 3) % 3) * HX + t0 % 3) * CK;
                const int off1 = (((t1 / 9) * HY + (t1 / 3) % 3) * HX + t1 % 3) * CK;
                const float* ap = abase + (hi ? off1 : off0);
#pragma unroll
                for (int r = 0; r < TY; ++r) {
                    const float4 a = *reinterpret_cast<const float4*>(ap + r * (HX * CK));
#pragma unroll
                    for (int nn = 0; nn < NREP; ++nn) {
                        acc[r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bcur[nn].x, acc[r][nn], 0, 0, 0);
                        acc[r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bcur[nn].y, acc[r][nn], 0, 0, 0);
                        acc[r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bcur[nn].z, acc[r][nn], 0, 0, 0);
                        acc[r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bcur[nn].w, acc[r][nn], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn) bcur[nn] = bnext[nn];
            }
        }
    }

    // epilogue.  C/D layout of 16x16x4: col (N = cout) = lane & 15, row (M = voxel x) = 4 * (lane >> 4) + reg
    const int z = z0 + wave;
    if (z >= p.D) return;
#pragma unroll
    for (int nn = 0; nn < NREP; ++nn) {
        const int co = (nt0 + nn) * 16 + i;
        if (co >= p.Cout) continue;
        const float b = p.bias ? p.bias[co] : 0.f;
        float* dst; int Cd, cd;
        if (co < p.Cs1) { dst = p.out1; Cd = p.Cs1; cd = co; } else { dst = p.out2; Cd = p.Cs2; cd = co - p.Cs1; }
#pragma unroll
        for (int r = 0; r < TY; ++r) {
            const int y = y0 + r;
            if (y >= p.H) continue;
            const long long rowbase = (((long long)n * p.D + z) * p.H + y) * p.W;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int x = x0 + 4 * g + reg;
                if (x < p.W) dst[(rowbase + x) * Cd + cd] = da_act(acc[r][nn][reg] + b, p.slope);
            }
        }
    }
}

// packed B operand: wp[chunk][step][ntile][lane][m]
__global__ void pack_fwd_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout,
                                        int CK, int NSTEPS, int NTpad, int flipped, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(idx & 3); const int lane = (int)((idx >> 2) & 63);
        long long rest = idx >> 8;
        const int nt = (int)(rest % NTpad); rest /= NTpad;
        const int s = (int)(rest % NSTEPS); const int ch = (int)(rest / NSTEPS);
        const int g = lane >> 4, j = lane & 15;
        int tap, c4;
        if (CK == 16) { tap = s; c4 = g; } else { const int qd = s * 4 + g; tap = qd >> 1; c4 = qd & 1; }
        const int cin = ch * CK + c4 * 4 + m, cout = nt * 16 + j;
        float v = 0.f;
        if (tap < 27 && cout < Cout && cin < Cin)
            v = flipped ? w[((size_t)(26 - tap) * Cout + cout) * Cin + cin] : w[((size_t)tap * Cin + cin) * Cout + cout];
        wp[idx] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------
struct WgP {
    const float* in1; const float* in2; int C1, C2;
    const float* dy; float* partial;
    int N, D, H, W, Cout, ntz, nty, ntx, ntiles, tiles_per_slab, O;
};

template <int CK, int NREP>
__global__ void __launch_bounds__(256, 2) conv3_mfma_wgrad_kernel(WgP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TZ = 2, HZ = TZ + 2, TVOX = TZ * TY * TX;
    constexpr int CG = NREP * 16;
    constexpr int TPW = (CK == 16) ? 7 : 4;                     // tap slots per wave (CK = 8: tap PAIRS, 14 in total)
    constexpr bool SWZ = (CG % 32) == 0;
    float* ldsA = lds;
    float* ldsY = lds + HZ * HY * HX * CK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4;
    const int ch = blockIdx.y, cg = blockIdx.z;
    const int cbase = ch * CK;
    const float* src; int Cs, choff;
    if (cbase < p.C1) { src = p.in1; Cs = p.C1; choff = cbase; } else { src = p.in2; Cs = p.C2; choff = cbase - p.C1; }

    // per-lane A offsets (floats) for this wave's tap slots
    int offA[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        int tap;
        if (CK == 16) tap = wave + 4 * k; else tap = 2 * (wave + 4 * k) + (i >> 3);
        if (tap > 26) tap = 26;                               // garbage slot, never written out
        offA[k] = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * CK + ((CK == 16) ? i : (i & 7));
    }
    f32x4 acc[TPW][NREP];
#pragma unroll
    for (int k = 0; k < TPW; ++k)
#pragma unroll
        for (int nn = 0; nn < NREP; ++nn) acc[k][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tile_begin = blockIdx.x * p.tiles_per_slab;
    int tile_end = tile_begin + p.tiles_per_slab; if (tile_end > p.ntiles) tile_end = p.ntiles;
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        int t = tile;
        const int tx = t % p.ntx; t /= p.ntx;
        const int ty = t % p.nty; t /= p.nty;
        const int tz = t % p.ntz; const int n = t / p.ntz;
        const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
        __syncthreads();
        stage_tile<CK, HZ>(ldsA, src, Cs, choff, n, z0, y0, x0, p.D, p.H, p.W);
        // dY tile [TVOX][CG] (channel halves XOR-swizzled by voxel parity when CG % 32 == 0)
        {
            constexpr int Q = CG / 4;
#pragma unroll 4
            for (int idx = threadIdx.x; idx < TVOX * Q; idx += 256) {
                const int c4 = idx % Q; const int v = idx / Q;
                const int x = x0 + (v & 15), y = y0 + ((v >> 4) & 7), z = z0 + (v >> 7);
                const int co = cg * CG + c4 * 4;
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (z < p.D && y < p.H && x < p.W && co < p.Cout)
                    val = *reinterpret_cast<const float4*>(p.dy + ((((long long)n * p.D + z) * p.H + y) * p.W + x) * p.Cout + co);
                int c = c4 * 4;
                if (SWZ) c ^= (v & 1) << 4;
                *reinterpret_cast<float4*>(ldsY + v * CG + c) = val;
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int s = 0; s < TVOX / 4; ++s) {
            const int v = 4 * s + g;
            const int vx = v & 15, vy = (v >> 4) & 7, vz = v >> 7;
            float b[NREP];
#pragma unroll
            for (int nn = 0; nn < NREP; ++nn) {
                int c = nn * 16;
                if (SWZ) c ^= (v & 1) << 4;
                b[nn] = ldsY[v * CG + c + i];
            }
            const float* ap = ldsA + ((vz * HY + vy) * HX + vx) * CK;
#pragma unroll
            for (int k = 0; k < TPW; ++k) {
                const float a = ap[offA[k]];
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn)
                    acc[k][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nn], acc[k][nn], 0, 0, 0);
            }
        }
    }
    // write this slab's partial dW[tap][cin][cout]; rows (M) = 4*g + reg, cols (N) = i
    float* part = p.partial + (size_t)blockIdx.x * p.O;
    const int Cin = p.C1 + p.C2;
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
#pragma unroll
        for (int nn = 0; nn < NREP; ++nn) {
            const int co = cg * CG + nn * 16 + i;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 4 * g + reg;
                int tap, ci;
                if (CK == 16) { tap = wave + 4 * k; ci = row; } else { tap = 2 * (wave + 4 * k) + (row >> 3); ci = row & 7; }
                if (tap < 27 && co < p.Cout && (CK == 16 || wave + 4 * k < 14))
                    part[((size_t)tap * Cin + cbase + ci) * p.Cout + co] = acc[k][nn][reg];
            }
        }
    }
}

__global__ void slab_reduce_kernel(const float* __restrict__ partial, int nparts, int O, float* __restrict__ out) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < O; o += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int b = 0; b < nparts; ++b) s += (double)partial[(size_t)b * O + o];
        out[o] = (float)s;
    }
}

static int pick_ck(int C1, int C2) {
    const int Cin = C1 + C2;
    if (Cin % 16 == 0 && C1 % 16 == 0) return 16;
    if (Cin % 8 == 0 && C1 % 8 == 0) return 8;
    return 0;
}
// N-tiles per workgroup: 3 for Cout = 48 (dgrad of the 48->16 layer), otherwise <= 2 so that 8*NREP*4 accumulators +
// fragments stay under 256 VGPRs (two workgroups per CU); more couts go to blockIdx.y.
static int pick_nrep(int NT) { return NT <= 3 ? NT : (NT % 2 == 0 ? 2 : (NT % 3 == 0 ? 3 : 2)); }

static size_t packed_bytes(int Cin, int Cout, int CK) {
    const int NT = (Cout + 15) / 16, NREP = pick_nrep(NT);
    const int NTpad = (NT + NREP - 1) / NREP * NREP;
    const int NSTEPS = (27 * CK + 15) / 16;
    return da_align((size_t)(Cin / CK) * NSTEPS * NTpad * 256 * sizeof(float));
}

struct WgPlan { int CK, NREP, ngroups, nchunks, ntz, nty, ntx, ntiles, nslabs, tps; size_t partial_bytes; };
static WgPlan wgrad_plan(int N, int D, int H, int W, int C1, int C2, int Cout) {
    WgPlan q;
    q.CK = pick_ck(C1, C2);
    const int NT = (Cout + 15) / 16;
    q.NREP = NT >= 2 ? 2 : 1;
    q.ngroups = (NT + q.NREP - 1) / q.NREP;
    q.nchunks = q.CK ? (C1 + C2) / q.CK : 1;
    q.ntz = (D + 1) / 2; q.nty = (H + TY - 1) / TY; q.ntx = (W + TX - 1) / TX;
    q.ntiles = N * q.ntz * q.nty * q.ntx;
    const size_t O = (size_t)27 * (C1 + C2) * Cout;
    long long slabs = 1024 / (q.nchunks * q.ngroups); if (slabs < 1) slabs = 1;
    const long long cap = (long long)((96ull << 20) / (O * 4)); if (slabs > cap) slabs = cap < 1 ? 1 : cap;
    if (slabs > q.ntiles) slabs = q.ntiles;
    q.tps = (int)da_cdiv(q.ntiles, slabs);
    q.nslabs = (int)da_cdiv(q.ntiles, q.tps);
    q.partial_bytes = da_align((size_t)q.nslabs * O * sizeof(float));
    return q;
}

}  // namespace

size_t da_conv3_mfma_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int stride) {
    if (stride != 1) return 0;
    size_t pk = 0;
    if (Cin % 8 == 0) { const size_t a = packed_bytes(Cin, Cout, Cin % 16 == 0 ? 16 : 8); if (a > pk) pk = a; const size_t b = packed_bytes(Cin, Cout, 8); if (b > pk) pk = b; }
    if (Cout % 8 == 0) { const size_t a = packed_bytes(Cout, Cin, Cout % 16 == 0 ? 16 : 8); if (a > pk) pk = a; }
    size_t part = 0;
    if (Cin % 8 == 0) part = wgrad_plan(N, D, H, W, Cin, 0, Cout).partial_bytes;
    return pk + part;
}

bool da_conv3_mfma_fwd_supported(int C1, int C2, int Cout, int stride) {
    if (stride != 1) return false;
    if (pick_ck(C1, C2) == 0) return false;
    if (Cout < 8) return false;            // Cout = 3 (flow) wastes 13/16 of every MFMA: direct kernel is faster
    return true;
}

template <int CK, int NREP>
static int launch_fwd_mfma(const FwdP& p, int gy, hipStream_t st) {
    const size_t shm = (size_t)6 * HY * HX * CK * sizeof(float);
    auto kern = conv3_mfma_fwd_kernel<CK, NREP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.ntiles, gy), dim3(256), shm, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}

int da_conv3_mfma_fwd(const float* in1, int C1, const float* in2, int C2, const float* w_tio, int w_is_flipped_tr,
                      const float* bias, float* out1, int Cs1, float* out2, int Cs2,
                      int N, int D, int H, int W, int Cout, int stride, float slope,
                      void* ws, size_t ws_bytes, hipStream_t st) {
    (void)stride;
    const int Cin = C1 + C2;
    const int CK = pick_ck(C1, C2);
    if (!CK) return DA_ERR_UNSUPPORTED;
    const int NT = (Cout + 15) / 16, NREP = pick_nrep(NT);
    const int gy = (NT + NREP - 1) / NREP, NTpad = gy * NREP;
    const int NSTEPS = (27 * CK + 15) / 16;
    const size_t pk = packed_bytes(Cin, Cout, CK);
    if (ws_bytes < pk) return DA_ERR_WS_SMALL;
    float* wp = (float*)ws;
    const long long total = (long long)(Cin / CK) * NSTEPS * NTpad * 256;
    hipLaunchKernelGGL(pack_fwd_weights_kernel, dim3(da_grid(total, 256, 1024)), dim3(256), 0, st, w_tio, wp, Cin, Cout, CK, NSTEPS, NTpad, w_is_flipped_tr, total);
    DA_LAUNCH_CHECK();
    FwdP p;
    p.in1 = in1; p.in2 = in2; p.C1 = C1; p.C2 = C2; p.wp = wp; p.bias = bias;
    p.out1 = out1; p.out2 = out2; p.Cs1 = Cs1; p.Cs2 = Cs2;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cout = Cout; p.NT = NTpad;
    p.ntz = (D + 3) / 4; p.nty = (H + TY - 1) / TY; p.ntx = (W + TX - 1) / TX;
    p.ntiles = N * p.ntz * p.nty * p.ntx; p.slope = slope;
#define DA_FWD_CASE(ck, nr) if (CK == ck && NREP == nr) return launch_fwd_mfma<ck, nr>(p, gy, st)
    DA_FWD_CASE(16, 1); DA_FWD_CASE(16, 2); DA_FWD_CASE(16, 3); DA_FWD_CASE(16, 4);
    DA_FWD_CASE(8, 1); DA_FWD_CASE(8, 2); DA_FWD_CASE(8, 3); DA_FWD_CASE(8, 4);
#undef DA_FWD_CASE
    return DA_ERR_UNSUPPORTED;
}

bool da_conv3_mfma_wgrad_supported(int C1, int C2, int Cout, int stride) {
    if (stride != 1) return false;
    if (pick_ck(C1, C2) == 0) return false;
    if (Cout % 4 != 0 || Cout < 8) return false;
    return true;
}

template <int CK, int NREP>
static int launch_wgrad_mfma(const WgP& p, const WgPlan& q, hipStream_t st) {
    const size_t shm = (size_t)(4 * HY * HX * CK + 2 * TY * TX * NREP * 16) * sizeof(float);
    auto kern = conv3_mfma_wgrad_kernel<CK, NREP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(q.nslabs, q.nchunks, q.ngroups), dim3(256), shm, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}

int da_conv3_mfma_wgrad(const float* in1, int C1, const float* in2, int C2, const float* dy, float* dw_tio,
                        int N, int D, int H, int W, int Cout, int stride, void* ws, size_t ws_bytes, hipStream_t st) {
    (void)stride;
    const WgPlan q = wgrad_plan(N, D, H, W, C1, C2, Cout);
    if (!q.CK) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < q.partial_bytes) return DA_ERR_WS_SMALL;
    WgP p;
    p.in1 = in1; p.in2 = in2; p.C1 = C1; p.C2 = C2; p.dy = dy; p.partial = (float*)ws;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cout = Cout;
    p.ntz = q.ntz; p.nty = q.nty; p.ntx = q.ntx; p.ntiles = q.ntiles; p.tiles_per_slab = q.tps;
    p.O = 27 * (C1 + C2) * Cout;
    int rc = DA_ERR_UNSUPPORTED;
    if (q.CK == 16 && q.NREP == 1) rc = launch_wgrad_mfma<16, 1>(p, q, st);
    else if (q.CK == 16 && q.NREP == 2) rc = launch_wgrad_mfma<16, 2>(p, q, st);
    else if (q.CK == 8 && q.NREP == 1) rc = launch_wgrad_mfma<8, 1>(p, q, st);
    else if (q.CK == 8 && q.NREP == 2) rc = launch_wgrad_mfma<8, 2>(p, q, st);
    if (rc) return rc;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(da_grid(p.O, 256)), dim3(256), 0, st, p.partial, q.nslabs, p.O, dw_tio);
    DA_LAUNCH_CHECK();
    return 0;
}
