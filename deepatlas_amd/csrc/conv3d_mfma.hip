// Implicit-GEMM 3x3x3 convolution on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32).
// Template switches on the same kernels: BF = opt-in bf16 matrix mode (bf16 LDS tiles / packed weights, v_mfma_f32_16x16x32_bf16 or
// 16x16x16 + ds_read_b64_tr_b16 for the weight gradient, fp32 accumulate; da_set_matrix_bf16), PRO = input prologue (the staged
// tensor is a raw producer output whose BatchNorm + activation is applied on the way into LDS; da_conv3d_k3_fwd_pro / _wgrad_pro).
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
#include "split_f16.h"      // da_split2, da_absmax4, da_wave_max_nonneg, da_scale_exp, da_pow2: the two-term fp16 split of the SP kernels

#include "conv3d_stage.h"
#ifndef DA_BF16_SMAP
#ifndef DA_BF16_SMAP2
#define DA_BF16_SMAP2 0  // ... also in the two-N-tile kernels when the tensors are bf16 (raw staging: half the parked registers): 96 -> 32 forward 0.316 -> 0.308 ms but + statistics 0.262 -> 0.317; off
#endif
#define DA_BF16_SMAP 1   // bf16 matrix-mode forward kernels: staging offsets from the per-thread halo map (0: the cursor; A/B builds)
#endif
#ifndef DA_SP_LB
#define DA_SP_LB 1   // split mode, one N-tile: K-steps of weight-fragment lookahead
#endif
#ifndef DA_RPB4
#define DA_RPB4 0   // split mode, one N-tile: row blocks of four instead of two (no gain measured: 48 -> 16 forward 1.60 -> 1.62 ms, +16 registers)
#endif
#ifndef DA_PIN
#define DA_PIN 1   // pin the m-outer MFMA order (keeps hipcc from chaining 4 dependent MFMAs on one accumulator)
#endif

namespace {


// BF: bf16 matrix mode (da_set_matrix_bf16): tensors stay fp32 in HBM, the staged tile and the packed weights are bf16 and one
// v_mfma_f32_16x16x16_bf16 (fp32 accumulate) replaces the four v_mfma_f32_16x16x4_f32 of a K-step -- same lane <-> (voxel, cin)
// mapping, so everything around the K loop is shared.  The kernel is then bound by HBM / LDS instead of the matrix pipe.
// DYN (experiment, DA_DYN_TILES=1): work-stealing tile walk.  Instead of a static share of its XCD's tile range a workgroup draws the
// next position from that XCD's counter (one atomic per tile by thread 0, two tiles ahead, handed to the other waves through a
// 4-entry LDS ring), so a kernel whose workgroups start at different times -- behind another persistent kernel on the other stream
// -- still finishes together.  Not for the STATS variant: its per-workgroup partial sums would then depend on the draw order.
// SP: split mode (da_set_matrix_mode(2)) -- fp32 products on the fp16 matrix pipe (da_split2).  Both operands are scaled by a power of two
// and split into two fp16 terms (activations while they are staged, at the scale of their (tile, channel chunk); weights in the pack kernel,
// at the scale of their channel chunk) and a K-step of an (M-tile, N-tile) pair is three v_mfma_f32_16x16x32_f16: a.h b.l + a.l b.h + a.h b.h,
// small terms first, fp32 accumulate -- 3/16 of the fp32 matrix instructions' pipe time.  The accumulators of an output tile live in the
// unit 2^E of the item being accumulated (E = activation exponent + weight exponent): at the start of every item they are multiplied by
// 2^(E - E_previous) (exact), in the epilogue by 2^-E.  E of a later chunk is capped at 40 above the smallest E the tile has seen, so the
// rescaled sums cannot overflow (a chunk 2^40 below its neighbours does not reach the fp32 sum anyway).  LDS holds the two planes (CK = 8:
// 2 x 17 KB); fragments of the next two rows are read while the current two rows' 6 MFMAs issue.
template <int CK, int NREP, bool MASKED = false, int STATS = 0, bool BF = false, bool PRO = false, bool DYN = false, bool SP = false, int S2F = 0, bool PAIR = false, bool HB = false, int WPE = 2>   // MASKED: sparse tap sets (stride-2 via space-to-depth); STATS: 1 BN partial sums of the output, 2 (data gradient) BatchNorm-BACKWARD sums of the layer that produced the input this gradient belongs to; PRO: input prologue; S2F: 1 virtual space-to-depth input, 2 depth-to-space stores; HB: bf16 activation storage; WPE: waves per SIMD the register allocation must leave room for
__global__ void __launch_bounds__(256, WPE) conv3_mfma_fwd_kernel(FwdP p) {
    static_assert(!HB || (BF && !SP && !DYN), "bf16 activation storage: bf16 matrix mode only");
    static_assert(STATS != 2 || (SP && NREP == 1 && !PRO), "BatchNorm-backward sums: the split mode's one-N-tile data gradient");
    static_assert(S2F == 0 || MASKED, "fused space-to-depth addressing belongs to the tap-masked (stride-2) variants");
    // PAIR (split mode, one N-tile): two consecutive 8-channel chunks share every 64-byte sector of their input.  Staged one work item apart
    // the second one misses L2 (the launch turns its L2 over in about one item time): FETCH_SIZE 1.8x the algorithmic bytes.  With PAIR the
    // loads of BOTH chunks are issued together during the odd item of a pair; the second chunk's data waits in registers (pre2) through the
    // even item.  The item loop is unrolled by two (lambda instantiated per phase) so that the even phase contains no load instructions.
    static_assert(!PAIR || (SP && NREP == 1 && !PRO), "paired staging: split mode, one N-tile, no input prologue");
    static_assert(!DYN || (!STATS && !MASKED && !PRO), "dynamic tile walk: plain forward / data-gradient variants only");
    static_assert(!SP || (BF && !MASKED && !DYN && CK == 8), "split mode: dense bf16 K = 32 kernels on 8-channel chunks");
    constexpr int NP = SP ? 2 : 1;                                  // operand planes
    constexpr bool RAW = HB && BF && !SP && !PRO && !MASKED && S2F == 0;       // bf16 tensors copied straight into the bf16 LDS image (da_buf_loadq)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // K32: the dense bf16 kernels use v_mfma_f32_16x16x32_bf16 (K = 32 = two taps x 16 cin, or four taps x 8 cin; 16 cycles per
    // SIMD for twice the K of the 16x16x16 form, which gfx950 issues at ~32 cycles); the sparse-tap variant keeps one tap per step.
    constexpr bool K32 = BF && !MASKED;
    using AElem = std::conditional_t<BF, short, float>;
    using Frag = std::conditional_t<SP, f16x8, std::conditional_t<K32, bf16x8, std::conditional_t<BF, s16x4, f32x4>>>;   // A / B fragment: 4 (8) consecutive cin of one voxel / one cout
    constexpr int EB = BF ? 2 : 4;                               // bytes per staged element
    constexpr int TZ = 4, HZ = TZ + 2;
    constexpr int NSTEPS = K32 ? (CK == 16 ? 14 : 7) : (27 * CK + 15) / 16;          // 27 (CK = 16) | 14 (CK = 8: two taps per K-step); K32: 14 | 7
    constexpr int NIT = StageGeom<CK, HZ>::NIT;
    // staging iterations whose loads are issued one work item ahead and parked in VGPRs during the MFMA phase; the
    // rest (register budget: 8*NREP*4 accumulators must leave two workgroups per CU) are fetched after the barrier
    constexpr int PRE = MASKED ? (NREP == 1 ? NIT : 4) : ((NREP <= 2) ? NIT : (NREP == 3 ? (NIT < 4 ? NIT : 4) : 0));
    static_assert(!PRO || (PRE == NIT && !MASKED), "the prologue variant keeps the whole next tile in registers");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4;
    const int nt0 = blockIdx.y * NREP;
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
    const int nchunks = (p.C1 + p.C2) / CK;
    // persistent: this workgroup walks its share of the brick-ordered tile list; work item = (tile, channel chunk)
    const TileWalk tw = tile_walk(p.ntiles);
    const int nitems = DYN ? 0x7FFFFFFF : tw.cnt * nchunks;
    if (!DYN && nitems <= 0) {
        if (STATS) for (int c = threadIdx.x; c < NREP * 16; c += 256) { const int co = blockIdx.y * NREP * 16 + c; if (co < p.Cout) { p.stats_partial[((size_t)blockIdx.x * 2) * p.Cout + co] = 0.0; p.stats_partial[((size_t)blockIdx.x * 2 + 1) * p.Cout + co] = 0.0; } }
        return;
    }
    // DYN: this workgroup's XCD range [xlo, xhi) of the brick order, its counter, and the ring of drawn positions.  (Out-of-range
    // buffer ATOMICS fault on gfx950 -- tools/ubench/buffer_atomic_oob.hip -- so the draw sits in an exec-mask branch of thread 0.)
    int xlo = 0, xhi = 0;
    int* sp = reinterpret_cast<int*>(reinterpret_cast<char*>(lds) + (size_t)StageGeom<CK, HZ>::TOTAL * 4 * EB * NP);
    int* ctr = nullptr;
    if constexpr (DYN) {
        const int G = gridDim.x, X = (G % 8 == 0) ? 8 : 1, xcd = blockIdx.x % X;
        xlo = (int)((long long)p.ntiles * xcd / X); xhi = (int)((long long)p.ntiles * (xcd + 1) / X);
        ctr = p.dyn_ctr + blockIdx.y * 8 + xcd;
        if (threadIdx.x == 0) { sp[0] = xlo + atomicAdd(ctr, 1); sp[1] = xlo + atomicAdd(ctr, 1); }
        __syncthreads();
        if (sp[0] >= xhi) return;
    }
    auto tile_pos = [&](int k) -> int {
        if constexpr (DYN) return __builtin_amdgcn_readfirstlane(sp[k & 3]);
        else return tw.lo + k * tw.J;
    };

    // work item = (k-th tile of this workgroup's walk, channel chunk): the current and the next item's coordinates are kept in SGPRs
    int cK = 0, cCh = 0, cN, cZ, cY, cX, nK, nCh, nN, nZ, nY, nX;
    auto tile_at = [&](int k, int& n, int& z0, int& y0, int& x0) {
        int pos = tile_pos(k); pos = pos < p.ntiles ? pos : p.ntiles - 1;               // (past the walk: any valid entry; never used)
        const int4 t = p.tiles[__builtin_amdgcn_readfirstlane(pos)];
        n = t.x; z0 = t.y; y0 = t.z; x0 = t.w;
    };
    auto advance = [&]() { nCh = cCh + 1; nK = cK; if (nCh == nchunks) { nCh = 0; nK = cK + 1; } tile_at(nK, nN, nZ, nY, nX); };
    tile_at(0, cN, cZ, cY, cX); advance();
    auto item_coords = [&](int which, int& n, int& z0, int& y0, int& x0, int& ch) {      // which: 0 = current item, 1 = next item
        if (which) { n = nN; z0 = nZ; y0 = nY; x0 = nX; ch = nCh; } else { n = cN; z0 = cZ; y0 = cY; x0 = cX; ch = cCh; }
    };
    auto issue_stage = [&](int item, float4* pre) {          // iterations [0, PRE)
        int n, z0, y0, x0, ch;
        item_coords(item, n, z0, y0, x0, ch);
        const int cbase = ch * CK;
        if constexpr (S2F == 1) stage_load_s2d<CK, HZ, 0, PRE, HB>(pre, p.in1, p.s2in, cbase, n, z0, y0, x0, p.D, p.H, p.W);
        else if (cbase < p.C1) stage_load<CK, HZ, 0, PRE, HB, RAW>(pre, p.in1, p.C1, cbase, n, z0, y0, x0, p.D, p.H, p.W);
        else stage_load<CK, HZ, 0, PRE, HB, RAW>(pre, p.in2, p.C2, cbase - p.C1, n, z0, y0, x0, p.D, p.H, p.W);
    };
    auto stage_rest = [&](int item) {                         // iterations [PRE, NIT): global -> LDS, in <= 3 batches
        if constexpr (PRE < NIT) {
            int n, z0, y0, x0, ch;
            item_coords(item, n, z0, y0, x0, ch);
            const int cbase = ch * CK;
            const float* src = (cbase < p.C1) ? p.in1 : p.in2;
            const int Cs = (cbase < p.C1) ? p.C1 : p.C2, choff = (cbase < p.C1) ? cbase : cbase - p.C1;
            constexpr int R = NIT - PRE, B1 = PRE + (R + 2) / 3, B2 = PRE + 2 * ((R + 2) / 3) < NIT ? PRE + 2 * ((R + 2) / 3) : NIT;
            if constexpr (S2F == 1) {
                {
                    { float4 tmp[B1 - PRE]; stage_load_s2d<CK, HZ, PRE, B1, HB>(tmp, p.in1, p.s2in, cbase, n, z0, y0, x0, p.D, p.H, p.W); stage_write<CK, HZ, PRE, B1, BF, SP>(lds, tmp); }
                    if constexpr (B2 > B1) { float4 tmp[B2 - B1]; stage_load_s2d<CK, HZ, B1, B2, HB>(tmp, p.in1, p.s2in, cbase, n, z0, y0, x0, p.D, p.H, p.W); stage_write<CK, HZ, B1, B2, BF, SP>(lds, tmp); }
                    if constexpr (NIT > B2) { float4 tmp[NIT - B2]; stage_load_s2d<CK, HZ, B2, NIT, HB>(tmp, p.in1, p.s2in, cbase, n, z0, y0, x0, p.D, p.H, p.W); stage_write<CK, HZ, B2, NIT, BF, SP>(lds, tmp); }
                }
            } else {
            { float4 tmp[B1 - PRE]; stage_load<CK, HZ, PRE, B1, HB, RAW>(tmp, src, Cs, choff, n, z0, y0, x0, p.D, p.H, p.W); stage_write<CK, HZ, PRE, B1, BF, SP, 0, RAW>(lds, tmp); }
            if constexpr (B2 > B1) { float4 tmp[B2 - B1]; stage_load<CK, HZ, B1, B2, HB, RAW>(tmp, src, Cs, choff, n, z0, y0, x0, p.D, p.H, p.W); stage_write<CK, HZ, B1, B2, BF, SP, 0, RAW>(lds, tmp); }
            if constexpr (NIT > B2) { float4 tmp[NIT - B2]; stage_load<CK, HZ, B2, NIT, HB, RAW>(tmp, src, Cs, choff, n, z0, y0, x0, p.D, p.H, p.W); stage_write<CK, HZ, B2, NIT, BF, SP, 0, RAW>(lds, tmp); }
            }
        }
    };

    f32x4 acc[TY][NREP];
#pragma unroll
    for (int r = 0; r < TY; ++r)
#pragma unroll
        for (int nn = 0; nn < NREP; ++nn) acc[r][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float st1[STATS ? NREP : 1][4], st2[STATS ? NREP : 1][4];   // per-lane BN partial sums of this lane's 4 couts (after the transpose)
#pragma unroll
    for (int nn = 0; nn < (STATS ? NREP : 1); ++nn)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[nn][j] = 0.f; st2[nn][j] = 0.f; }
    // BN partial sums: per-lane fp32 sums are folded every 2 tiles (<= 16 values per lane and channel) into per-channel
    // DOUBLE accumulators held by the first NREP*16 threads (wave shuffles over q/g, then the 4 waves through a 2 KiB LDS
    // strip behind the tile; everything past the per-lane sums is double), so E[x^2] - mean^2 keeps the accuracy of the stand-alone statistics pass.
    double dsum1 = 0.0, dsum2 = 0.0;
    double* sred = reinterpret_cast<double*>(reinterpret_cast<char*>(lds) + (size_t)StageGeom<CK, HZ>::TOTAL * 4 * EB * NP);      // [wave][2][NREP*16] doubles
    int tiles_done = 0;
    auto stats_flush = [&]() {
        if constexpr (STATS) {
#pragma unroll
            for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double a = (double)st1[nn][j], b = (double)st2[nn][j];     // lane-to-lane tree in double
                    a += __shfl_xor(a, 1); b += __shfl_xor(b, 1);
                    a += __shfl_xor(a, 2); b += __shfl_xor(b, 2);
                    a += __shfl_xor(a, 4); b += __shfl_xor(b, 4);
                    a += __shfl_xor(a, 8); b += __shfl_xor(b, 8);
                    if ((lane & 15) == 0) {
                        const int c = nn * 16 + 4 * (lane >> 4) + j;
                        sred[(wave * 2 + 0) * (NREP * 16) + c] = a;
                        sred[(wave * 2 + 1) * (NREP * 16) + c] = b;
                    }
                    st1[nn][j] = 0.f; st2[nn][j] = 0.f;
                }
            __syncthreads();
            if ((int)threadIdx.x < NREP * 16) {
                const int c = threadIdx.x;
#pragma unroll
                for (int w = 0; w < 4; ++w) { dsum1 += sred[(w * 2 + 0) * (NREP * 16) + c]; dsum2 += sred[(w * 2 + 1) * (NREP * 16) + c]; }
            }
            __syncthreads();
        }
    };
    float4 pre[PRE > 0 ? PRE : 1];
    float4 pre2[PAIR ? PRE : 1];          // PAIR: the second chunk of the pair staged during the previous odd item
    // PRO: scale / shift / slope of the channel quad this thread stages (256 % Q == 0: the quad is fixed per thread and chunk).
    // Unconditional loads from always-valid arrays (the host substitutes identity arrays for an input without a prologue).
    unsigned vm = 0;
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psf = make_float4(0.f, 0.f, 0.f, 0.f); float pslope = -1.f;
    auto load_pro = [&](int item) {
        if constexpr (PRO) {
            int n, z0, y0, x0, ch;
            item_coords(item, n, z0, y0, x0, ch);
            const int cbase = ch * CK;
            const bool first = cbase < p.C1;
            const int cofs = (first ? cbase : cbase - p.C1) + ((int)threadIdx.x % StageGeom<CK, HZ>::Q) * 4;
            psc = *reinterpret_cast<const float4*>((first ? p.ps1 : p.ps2) + cofs);
            psf = *reinterpret_cast<const float4*>((first ? p.pt1 : p.pt2) + cofs);
            pslope = first ? p.pslope1 : p.pslope2;
        }
    };
    // SP: scale bookkeeping, all wave-uniform.  The accumulators hold (true sum) x 2^Eacc; the item in LDS was staged at 2^(Ecur - its weight
    // exponent); Emin = smallest E of the current output tile so far.  The four waves' tile maxima meet in a 16-byte strip behind the tile.
    float* smax = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + (size_t)StageGeom<CK, HZ>::TOTAL * 4 * EB * NP + (STATS ? (size_t)4 * 2 * NREP * 16 * sizeof(double) : 0));
    int Ecur = 0, Eacc = 0, Emin = 0, Enext = 0;
    auto sp_publish = [&](const float4* q) {                 // before the barrier that retires the current tile
        const float m = da_wave_max_nonneg(stage_absmax<PRE>(q));
        if (lane == 0) smax[wave] = m;
    };
    auto sp_scale = [&](int chn) -> float {                  // after it: the activation scale of the tile about to be written (chunk chn); sets Enext
        const float4 mm = *reinterpret_cast<const float4*>(smax);
        const int mi = __builtin_amdgcn_readfirstlane(__float_as_int(fmaxf(fmaxf(mm.x, mm.y), fmaxf(mm.z, mm.w))));
        const int ew = p.wexp[chn];
        int E = da_scale_exp(__int_as_float(mi)) + ew;
        if (chn != 0) E = min(E, Emin + 40);
        // keep the accumulators' unit (Ecur: the item being accumulated) while the next chunk fits it -- staged up to 8x smaller than its own
        // maximum allows; see conv3_split_wgrad_kernel -- so that the per-item accumulator multiplies run only when the magnitude moves
        if (chn != 0 && E >= Ecur && E <= Ecur + 3 && !(p.ablate & 8)) E = Ecur;
        Emin = (chn == 0) ? E : min(Emin, E);
        Enext = E;
        return da_pow2(E - ew);
    };
    if constexpr (PRO) {
        int n, z0, y0, x0, ch;
        item_coords(0, n, z0, y0, x0, ch);
        const int cbase = ch * CK;
        if (cbase < p.C1) stage_load<CK, HZ, 0, PRE, HB>(pre, p.in1, p.C1, cbase, n, z0, y0, x0, p.D, p.H, p.W, &vm);
        else stage_load<CK, HZ, 0, PRE, HB>(pre, p.in2, p.C2, cbase - p.C1, n, z0, y0, x0, p.D, p.H, p.W, &vm);
        load_pro(0);
        if constexpr (SP) {
            stage_pro_apply<0, PRE>(pre, vm, psc, psf, pslope);
            sp_publish(pre); __syncthreads();
            const float s0 = sp_scale(0); Ecur = Eacc = Enext;
            stage_write<CK, HZ, 0, PRE, BF, SP>(lds, pre, s0);
        } else stage_write_pro<CK, HZ, 0, PRE, BF>(lds, pre, vm, psc, psf, pslope);
    } else {
    issue_stage(0, pre);
    if constexpr (SP) {
        sp_publish(pre); __syncthreads();
        const float s0 = sp_scale(0); Ecur = Eacc = Enext;
        stage_write<CK, HZ, 0, PRE, BF, SP>(lds, pre, s0);
    } else stage_write<CK, HZ, 0, PRE, BF, SP, 0, RAW>(lds, pre);
    stage_rest(0);
    if constexpr (PAIR) stage_load<CK, HZ, 0, PRE>(pre2, p.in1, p.C1, CK, cN, cZ, cY, cX, p.D, p.H, p.W);      // item 1 = chunk 1 of the first tile
    }
    __syncthreads();

    // K-step s: lane group g supplies (tap, cin quad) = CK16: (s, g) | CK8: (2s + (g>>1), g&1).
    const AElem* abase = K32 ? reinterpret_cast<const AElem*>(lds) + ((wave * HY) * HX + i) * CK + ((CK == 16) ? (g & 1) * 8 : 0)
                       : (CK == 16) ? reinterpret_cast<const AElem*>(lds) + ((wave * HY) * HX + i) * CK + g * 4
                                    : reinterpret_cast<const AElem*>(lds) + ((wave * HY) * HX + i) * CK + (g & 1) * 4;
    // SMAP: the next item's staging offsets from the per-thread halo map (7 VALU operations per load for an interior tile) instead of the
    // carry-stepping cursor with its per-load bounds checks (~20).  Split mode always; the bf16 matrix-mode kernels too when the whole next
    // tile is parked in registers (PRE == NIT): their 8 MFMAs per K-step leave the VALU as the busiest pipe (SQ counters: 7 VALU instructions
    // per MFMA with the cursor, profiles/r03_pmc_sq_conv3d_48to16.txt)
    constexpr bool SMAP = SP || (K32 && PRE == NIT && (NREP == 1 || (RAW && DA_BF16_SMAP2)) && DA_BF16_SMAP);      // (two N-tiles with fp32 tensors: measured 10 % slower with the map's 17 extra registers; raw bf16 staging parks half the registers)
    constexpr bool SMVO = SP && !(NREP == 2 && STATS);            // the launch-constant voxel offsets cost NIT registers
    StageMap<CK, HZ, SMVO> smap; if constexpr (SMAP) smap.init(p.H, p.W);
    const bool hi = (g >> 1) != 0;
    auto a_off = [&](int s) -> int { return (((s / 9) * HY + (s / 3) % 3) * HX + s % 3) * CK; };   // CK16, s = tap
    auto a_off8 = [&](int s) -> int {                                                              // CK8: two taps per step
        int tap = 2 * s + (hi ? 1 : 0); if (tap > 26) tap = 26;
        return (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * CK;
    };
    auto a_off32 = [&](int s) -> int {                                                             // K32: lane group g -> tap 2s + (g>>1) | 4s + g
        int tap = (CK == 16) ? 2 * s + (g >> 1) : 4 * s + g; if (tap > 26) tap = 26;             // (taps past 26 carry zero weights)
        return (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * CK;
    };
    int aoff32[SP ? NSTEPS : 1];                         // SP: per-lane tap offsets of the K-steps, computed once instead of once per item
    if constexpr (SP) {
#pragma unroll
        for (int t = 0; t < NSTEPS; ++t) aoff32[t] = a_off32(t);
    }

    // ---- vector-memory scheduling.  vmcnt retires IN ORDER on gfx9 and counts stores as well as loads, and the CU's vector
    // memory pipe is a FIFO shared by both resident workgroups.  A B fragment requested behind a 69 KB staging burst (or behind
    // the epilogue's stores) is therefore unusable until all of that has completed, which stalled every item by a memory
    // latency or two (ablation: staging loads +10 %, epilogue stores +9 % of the kernel time, not overlapped with anything).
    // So: (1) B fragments come through a buffer descriptor (scalar step offset, no VGPR address math) and are requested
    // LB K-steps ahead in a register ring that runs across item boundaries (the first LB steps of the next item are requested
    // before this item's epilogue stores); (2) the next item's staging loads are spread over the K-steps, at most one per step;
    // (3) nothing that touches vector memory sits inside a branch: invalid work (no next item, not the last chunk, ragged
    // lanes) is expressed as out-of-range buffer offsets, so hipcc's s_waitcnt vmcnt(N) stay exact instead of collapsing to 0.
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (unsigned)(nchunks * NSTEPS * p.NT * (K32 ? 1024 * NP : 256 * EB)), 0x00020000);
    auto wb = [&](int chunk, int step, int nn, int pl = 0) -> Frag {
        if constexpr (SP) return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, (unsigned)lane * 16u, (unsigned)((((chunk * NSTEPS + step) * p.NT + nt0 + nn) * 2 + pl) * 1024), 0));
        else if constexpr (K32) return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, (unsigned)lane * 16u, (unsigned)(((chunk * NSTEPS + step) * p.NT + nt0 + nn) * 1024), 0));
        else if constexpr (BF) return __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(rsw, (unsigned)lane * 8u, (unsigned)(((chunk * NSTEPS + step) * p.NT + nt0 + nn) * 512), 0));
        else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (unsigned)lane * 16u, (unsigned)(((chunk * NSTEPS + step) * p.NT + nt0 + nn) * 1024), 0));
    };
    // one K-step of one (M-tile, N-tile) pair
    auto mma_bf = [&](f32x4 c, const Frag& a, const Frag& b) -> f32x4 {
        if constexpr (SP) return __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c, 0, 0, 0);          // (weights as A, activations as B: see bvv)
        else if constexpr (K32) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
        else if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(b, a, c, 0, 0, 0);
        else return c;
    };
    constexpr int LB = SP ? (NREP == 1 ? DA_SP_LB : 1) : (NREP == 1) ? 4 : ((NREP == 2 && !STATS) ? 2 : 1);   // B lookahead in K-steps (the statistics variant of NREP = 2 would spill at 2; a split-mode K-step is 6 x 8 MFMAs long)
    constexpr int RB = LB + 1;                          // ring slots
    constexpr int TAIL = SP ? 2 : 5;                    // K-steps at the end of an item without staging loads (they must land before stage_write)
    constexpr int PRO_DELAY = SP ? 1 : 3;               // K-steps between a staging load and its prologue arithmetic (< TAIL)
    constexpr bool PRO_IN = PRO && NREP == 1;           // prologue arithmetic inside the K loop (one N-tile: registers to spare); else between the barriers
    static_assert(!PRO || NSTEPS - TAIL + PRO_DELAY <= NSTEPS, "prologue arithmetic must fall inside the K loop");
    Frag bq[RB][NREP][NP], nb[LB][NREP][NP];
    if constexpr (!MASKED) {
#pragma unroll
        for (int t = 0; t < LB; ++t)
#pragma unroll
            for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) nb[t][nn][pl] = wb(0, t, nn, pl);     // item 0 is chunk 0
    }
    // bias of this lane's 4 couts (constant for the whole launch).  The MFMAs take the WEIGHT fragment as their A operand and the activation
    // fragment as B (both have the same per-lane layout), so D = [cout][voxel]: lane (i, g) holds voxel x0 + i and the four consecutive
    // couts 4 g .. 4 g + 3 -- one 16-byte store per M-tile with no transpose in the epilogue.
    const int a4 = g;
    float bvv[NREP][4];
#pragma unroll
    for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int co = (nt0 + nn) * 16 + 4 * a4 + j; bvv[nn][j] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f; }

    const int prio_rank = (int)((blockIdx.x + gridDim.x * blockIdx.y) / 256u);
    // one work item; PH: 0 = unpaired, 1 = even item of a pair (no staging loads: the next item's tile is in pre2), 2 = odd item (loads both
    // chunks of the next pair).  Returns false when the walk is over (DYN).
    auto item_body = [&](int item, auto PHC) -> bool {
        constexpr int PH = decltype(PHC)::value;
        if (p.prio_ranks > 1) da_setprio((prio_rank + item) % p.prio_ranks);
        int n, z0, y0, x0, ch;
        item_coords(0, n, z0, y0, x0, ch);
        const bool has_next = PH == 1 ? true : DYN ? ((ch + 1 < nchunks) || tile_pos(cK + 1) < xhi) : (item + 1 < nitems);
        const bool last = (ch == nchunks - 1);
        if constexpr (SP) {          // bring the running sums into this item's unit (exact: a power of two; zero sums on a tile's first chunk)
            if (Ecur != Eacc) {
                if (ch != 0) {                                   // (a tile's first chunk starts from zero sums)
                    const float f = da_acc_factor(Ecur - Eacc);
#pragma unroll
                    for (int r = 0; r < TY; ++r)
#pragma unroll
                        for (int nn = 0; nn < NREP; ++nn) acc[r][nn] = acc[r][nn] * f;
                }
                Eacc = Ecur;
            }
        }
        if constexpr (DYN) {       // on a tile's first chunk: draw the position of the tile after next; published below, before the barriers
            if (threadIdx.x == 0 && ch == 0) sp[(cK + 2) & 3] = xlo + atomicAdd(ctr, 1);
        }

        if constexpr (MASKED) {
            // sparse tap set (a stride-2 conv expressed as a stride-1 conv over the space-to-depth input: a channel chunk
            // belongs to one input parity and only (1|2)^3 of the 27 taps are non-zero).  Staging-bound, so a plain loop.
            const Frag* wch = reinterpret_cast<const Frag*>(p.wp) + ((size_t)ch * NSTEPS * p.NT + nt0) * 64 + lane;
            if (has_next && !(p.ablate & 1)) issue_stage(1, pre);
            unsigned msk = p.masks[p.maskmode == 1 ? ch : (int)blockIdx.y];
            // software-pipelined over the live taps: the fragments of the NEXT tap (weights from global / L2, eight A rows from LDS) are
            // requested before the current tap's 32 * NREP MFMAs issue, so neither latency sits between two taps (the plain loop paid
            // both per tap: a chunk has only 1 .. 8 taps, there is nothing else to hide them behind).
            auto load_tap = [&](int sidx, Frag (&bb)[NREP], Frag (&aa)[TY]) {
                const AElem* ap = abase + (((sidx / 9) * HY + (sidx / 3) % 3) * HX + sidx % 3) * CK;
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn) bb[nn] = wch[((size_t)sidx * p.NT + nn) * 64];
#pragma unroll
                for (int r = 0; r < TY; ++r) aa[r] = *reinterpret_cast<const Frag*>(ap + r * (HX * CK));
            };
            if (msk) {
                Frag bb[NREP], aa[TY];
                { const int s0 = __builtin_ctz(msk); msk &= msk - 1; load_tap(s0, bb, aa); }
                while (true) {
                    const bool more = msk != 0;
                    Frag bn[NREP], an[TY];
                    if (more) { const int s1 = __builtin_ctz(msk); msk &= msk - 1; load_tap(s1, bn, an); }
                    if constexpr (BF) {
#pragma unroll
                        for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                            for (int r = 0; r < TY; ++r) acc[r][nn] = mma_bf(acc[r][nn], aa[r], bb[nn]);
                    } else {
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                                for (int r = 0; r < TY; ++r)
                                    acc[r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[nn][m], aa[r][m], acc[r][nn], 0, 0, 0);
                    }
                    if (!more) break;
#pragma unroll
                    for (int nn = 0; nn < NREP; ++nn) bb[nn] = bn[nn];
#pragma unroll
                    for (int r = 0; r < TY; ++r) aa[r] = an[r];
                }
            }
        } else {
        // CK = 16: 27 K-steps; CK = 8: 14 (two taps per step), fully unrolled.  Each K-step is split into two half-steps of 4
        // M-tiles; the A fragments of the NEXT half-step are read from LDS while the current half-step's 16*NREP MFMAs issue (a
        // 2 x 4-fragment double buffer = the same 32 VGPRs one full step of fragments needs), so no ds_read latency is exposed.
        // MFMA order: component m outermost, M-tile r innermost -> 4*NREP independent accumulators between two uses of the
        // same one (v_mfma_f32_16x16x4_f32: 32-cycle issue, 40-cycle dependent latency).
        constexpr int HALF = TY / 2;
        auto step_ptr = [&](int s) -> const AElem* { return SP ? abase + aoff32[s] : K32 ? abase + a_off32(s) : (CK == 16) ? abase + a_off(s) : abase + a_off8(s); };
        const int ch_next = (ch + 1 == nchunks) ? 0 : ch + 1;
#pragma unroll
        for (int t = 0; t < LB; ++t)
#pragma unroll
            for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) bq[t % RB][nn][pl] = nb[t][nn][pl];
        StageCursor<CK, HZ, HB, RAW> cur;                     // next item's staging loads: cursor (fp32 / bf16 kernels) ...
        typename StageMap<CK, HZ, SMVO>::Tile stile;                // ... or per-thread halo map (SP: registers to spare, ~60 % fewer instructions)
        __amdgpu_buffer_rsrc_t rsn;
        if constexpr (PRO) vm = 0;
        if constexpr (PRO_IN) load_pro(has_next ? 1 : 0);               // constants of the tile staged during this item
        {
            int n2, z2, y2, x2, ch2;
            item_coords(1, n2, z2, y2, x2, ch2);
            const int cbase = ch2 * CK;
            const bool first = cbase < p.C1;
            const int Csn = first ? p.C1 : p.C2;
            if constexpr (SP && PH == 1) { (void)Csn; }
            else if constexpr (SMAP) {
                const long long sample = (long long)p.D * p.H * p.W * Csn;
                rsn = da_rsrc_n<HB>(first ? p.in1 : p.in2, n2, sample);
                stile = smap.tile(z2, y2, x2, p.D, p.H, p.W, Csn, first ? cbase : cbase - p.C1, has_next && !(p.ablate & 1), (int)HbEl<HB>::ES);
            } else cur.init(first ? p.in1 : p.in2, Csn, first ? cbase : cbase - p.C1, n2, z2, y2, x2, p.D, p.H, p.W, has_next && !(p.ablate & 1));
        }
        Frag A0[HALF], A1[HALF];
        constexpr int PLANE_E = StageGeom<CK, HZ>::TOTAL * 4;             // elements per operand plane (SP)
        // SP: rows per block of MFMAs -- a block issues its products plane pair by plane pair over RPB x NREP accumulators, so two MFMAs on the same
        // accumulator are four apart (two apart, one N-tile and row pairs, cost 2 - 5 wait states per MFMA: its latency is two issue slots)
        constexpr int RPB = (SP && NREP == 1 && WPE == 2 && DA_RPB4) ? 4 : 2;
        Frag AC[NP][RPB];                                                   // SP: fragments of the current row block
        if constexpr (SP) {
            const AElem* ap = step_ptr(0);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int rr = 0; rr < RPB; ++rr) AC[pl][rr] = *reinterpret_cast<const Frag*>(ap + pl * PLANE_E + rr * (HX * CK));
        } else {
            const AElem* ap = step_ptr(0);
#pragma unroll
            for (int r = 0; r < HALF; ++r) A0[r] = *reinterpret_cast<const Frag*>(ap + r * (HX * CK));
        }
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            // weights of K-step s + LB (of this chunk, or of the next item's chunk)
#pragma unroll
            for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    if (s + LB < NSTEPS) bq[(s + LB) % RB][nn][pl] = wb(ch, s + LB, nn, pl);
                    else nb[s + LB - NSTEPS][nn][pl] = wb(ch_next, s + LB - NSTEPS, nn, pl);
                }
            // next item's staging loads scheduled on this step
#pragma unroll
            for (int j = 0; j < PRE; ++j)
                if (j * (NSTEPS - TAIL) / (PRE > 0 ? PRE : 1) == s) {
                    if constexpr (SP && PH == 1) { }
                    else if constexpr (SMAP) {
                        const unsigned so = smap.offset(stile, j);
                        pre[j] = da_buf_loadq<HB, RAW>(rsn, so);
                        if constexpr (PH == 2) pre2[j] = da_buf_load4(rsn, so == 0xFFFFFFFFu ? so : so + CK * 4u);      // the same voxel's next 8 channels
                        if constexpr (PRO) vm |= (so != 0xFFFFFFFFu ? 1u : 0u) << j;
                    } else { pre[j] = cur.next(); if constexpr (PRO) vm |= (cur.last_inb ? 1u : 0u) << j; }
                }
            if constexpr (PRO_IN) {   // the deferred BatchNorm + activation of a staged quad, PRO_DELAY K-steps after its load was issued:
                                      // by then it has landed, and the VALU work hides under the other wave's MFMAs instead of sitting between the barriers
#pragma unroll
                for (int j = 0; j < PRE; ++j)
                    if (j * (NSTEPS - TAIL) / (PRE > 0 ? PRE : 1) + PRO_DELAY == s) {
                        const bool ok = ((vm >> j) & 1u) != 0;
                        const float4 t = pre[j];
                        pre[j].x = ok ? da_act01(t.x * psc.x + psf.x, pslope) : 0.f; pre[j].y = ok ? da_act01(t.y * psc.y + psf.y, pslope) : 0.f;
                        pre[j].z = ok ? da_act01(t.z * psc.z + psf.z, pslope) : 0.f; pre[j].w = ok ? da_act01(t.w * psc.w + psf.w, pslope) : 0.f;
                    }
            }
            const AElem* ap = step_ptr(s);
            if constexpr (SP) {
                // row blocks: the two planes' fragments of the NEXT block (or of the next K-step's first block) are read while the 3 * RPB * NREP
                // MFMAs of the current block issue.  Products small terms first: (a, b) plane pairs (h, l) (l, h) (h, h).
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
                const AElem* anx = (s + 1 < NSTEPS) ? step_ptr(s + 1) : ap;
#pragma unroll
                for (int qd = 0; qd < TY / RPB; ++qd) {
                    Frag AN[NP][RPB];
                    const AElem* src = (qd + 1 < TY / RPB) ? ap + (RPB * qd + RPB) * (HX * CK) : anx;
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr) AN[pl][rr] = *reinterpret_cast<const Frag*>(src + pl * PLANE_E + rr * (HX * CK));
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                            for (int rr = 0; rr < RPB; ++rr)
                                acc[RPB * qd + rr][nn] = mma_bf(acc[RPB * qd + rr][nn], AC[PA[pr] % NP][rr], bq[s % RB][nn][PB[pr] % NP]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr) AC[pl][rr] = AN[pl][rr];
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < HALF; ++r) A1[r] = *reinterpret_cast<const Frag*>(ap + (HALF + r) * (HX * CK));
            if constexpr (BF) {
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                    for (int r = 0; r < HALF; ++r) acc[r][nn] = mma_bf(acc[r][nn], A0[r], bq[s % RB][nn][0]);
            } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                    for (int r = 0; r < HALF; ++r)
                        acc[r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[s % RB][nn][0][m], A0[r][m], acc[r][nn], 0, 0, 0);
                if (DA_PIN) __builtin_amdgcn_sched_barrier(0);
            }
            }
            {   // first half of the next K-step (the last step re-reads its own, harmlessly)
                const AElem* an = (s + 1 < NSTEPS) ? step_ptr(s + 1) : ap;
#pragma unroll
                for (int r = 0; r < HALF; ++r) A0[r] = *reinterpret_cast<const Frag*>(an + r * (HX * CK));
            }
            if constexpr (BF) {
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                    for (int r = 0; r < HALF; ++r) acc[HALF + r][nn] = mma_bf(acc[HALF + r][nn], A1[r], bq[s % RB][nn][0]);
            } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                    for (int r = 0; r < HALF; ++r)
                        acc[HALF + r][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[s % RB][nn][0][m], A1[r][m], acc[HALF + r][nn], 0, 0, 0);
                if (DA_PIN) __builtin_amdgcn_sched_barrier(0);
            }
            }
        }
        }

        // epilogue.  C/D layout of the 16x16 MFMAs: col (N) = lane & 15, row (M) = 4 * (lane >> 4) + reg.  With the weights as the A operand
        // M = cout and N = voxel x: a lane holds 1 voxel x 4 consecutive couts = one 16-byte store per M-tile, 1 KiB contiguous per wave
        // instruction for Cout = 16, and no transpose.  The arithmetic runs on the last chunk only; the STORES are issued on every item through a buffer
        // descriptor, with the offset out of range (dropped by the hardware) when this is not the last chunk or the lane is
        // outside the volume -- no vector-memory instruction sits in a branch (see above).
        {
            const int z = z0 + wave;
            const int x = x0 + i;
            const bool do_ep = last && !(p.ablate & 2);
            const float inv1 = SP ? da_pow2(-(Ecur / 2)) : 1.f, inv2 = SP ? da_pow2(-(Ecur - Ecur / 2)) : 1.f;
            // STATS == 2: the producer layer's raw output at this lane's voxels and channel quad (issued on every item, out of range -- no traffic --
            // unless this is a tile's last chunk: no vector-memory instruction in a branch), and its per-channel constants
            float4 yq[STATS == 2 ? TY : 1], bsc = make_float4(0.f, 0.f, 0.f, 0.f), bsf = bsc, bmu = bsc;
            if constexpr (STATS == 2) {
                const int cq = nt0 * 16 + 4 * a4;
                const bool cokq = do_ep && (cq + 3 < p.Cout) && z < p.D && x < p.W;
                const __amdgpu_buffer_rsrc_t ry = da_rsrc_n<false>(p.bst_y, n, (long long)p.D * p.H * p.W * p.Cs1);
#pragma unroll
                for (int r = 0; r < TY; ++r) {
                    const unsigned off = ((unsigned)((z * p.H + (y0 + r)) * p.W + x) * (unsigned)p.Cs1 + (unsigned)cq) * 4u;      // (unsigned: the guard bounds the byte offset below 2^32, not 2^31)
                    yq[r] = da_buf_load4(ry, (cokq && y0 + r < p.H) ? off : 0xFFFFFFFFu);
                }
                const __amdgpu_buffer_rsrc_t rp = da_rsrc(p.bst_par, (unsigned)(4 * p.Cs1 * 4));
                const unsigned po = (do_ep && cq + 3 < p.Cout) ? (unsigned)(cq * 4) : 0xFFFFFFFFu;
                bmu = da_buf_load4(rp, po);
                bsc = da_buf_load4(rp, po == 0xFFFFFFFFu ? po : po + (unsigned)(2 * p.Cs1 * 4));
                bsf = da_buf_load4(rp, po == 0xFFFFFFFFu ? po : po + (unsigned)(3 * p.Cs1 * 4));
            }
            if (do_ep) {
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn) {
#pragma unroll
                    for (int r = 0; r < TY; ++r) {
                        float t0 = acc[r][nn][0], t1 = acc[r][nn][1], t2 = acc[r][nn][2], t3 = acc[r][nn][3];      // couts co0..co0+3 of voxel (z, y0 + r, x)
                        if constexpr (SP) { t0 = t0 * inv1 * inv2; t1 = t1 * inv1 * inv2; t2 = t2 * inv1 * inv2; t3 = t3 * inv1 * inv2; }      // back to the true unit (two exact factors: |E| may exceed 127)
                        const float v0 = t0 + bvv[nn][0], v1 = t1 + bvv[nn][1], v2 = t2 + bvv[nn][2], v3 = t3 + bvv[nn][3];
                        if constexpr (STATS == 2) {
                            const float m = (z < p.D && y0 + r < p.H && x < p.W) ? 1.f : 0.f;
                            const float4 yv = yq[r];
                            const float d0 = m * v0 * da_act_grad(yv.x * bsc.x + bsf.x, p.bst_slope), d1 = m * v1 * da_act_grad(yv.y * bsc.y + bsf.y, p.bst_slope);
                            const float d2 = m * v2 * da_act_grad(yv.z * bsc.z + bsf.z, p.bst_slope), d3 = m * v3 * da_act_grad(yv.w * bsc.w + bsf.w, p.bst_slope);
                            st1[nn][0] += d0; st1[nn][1] += d1; st1[nn][2] += d2; st1[nn][3] += d3;
                            st2[nn][0] += d0 * (yv.x - bmu.x); st2[nn][1] += d1 * (yv.y - bmu.y); st2[nn][2] += d2 * (yv.z - bmu.z); st2[nn][3] += d3 * (yv.w - bmu.w);
                        } else if (STATS) {
                            const float m = (z < p.D && y0 + r < p.H && x < p.W) ? 1.f : 0.f;
                            st1[nn][0] += m * v0; st1[nn][1] += m * v1; st1[nn][2] += m * v2; st1[nn][3] += m * v3;
                            st2[nn][0] += m * v0 * v0; st2[nn][1] += m * v1 * v1; st2[nn][2] += m * v2 * v2; st2[nn][3] += m * v3 * v3;
                        }
                        acc[r][nn] = (f32x4){da_act(v0, p.slope), da_act(v1, p.slope), da_act(v2, p.slope), da_act(v3, p.slope)};
                        if constexpr (STATS) __builtin_amdgcn_sched_barrier(0);      // one M-tile at a time (interleaving all of them spills the statistics variants)
                    }
                }
            }
#pragma unroll
            for (int nn = 0; nn < NREP; ++nn) {
                const int cb = (nt0 + nn) * 16;                    // this N-tile lies entirely in out1 or in out2 (Cs1 % 16 == 0 when split)
                const bool first = cb < p.Cs1;
                float* dbase = first ? p.out1 : p.out2;
                const int Cd = first ? p.Cs1 : p.Cs2;
                const int cd = cb - (first ? 0 : p.Cs1) + 4 * a4;
                const bool cok = do_ep && (cb + 4 * a4 + 3 < p.Cout) && z < p.D && x < p.W;
                constexpr bool scattered = (S2F == 2);
                if constexpr (S2F == 2) {
                    {      // depth-to-space folded into the stores: N-tile cb = (parity, 16 channels) of the original-resolution tensor
                        const int rr = cb / p.s2out.cin, c0 = cb - rr * p.s2out.cin + 4 * a4;
                        const int zs = 2 * z + ((rr >> 2) & 1), xs = 2 * x + (rr & 1), ry = (rr >> 1) & 1;
                        const long long sample0 = (long long)p.s2out.D0 * p.s2out.H0 * p.s2out.W0 * p.s2out.cin;
                        const __amdgpu_buffer_rsrc_t r0 = da_rsrc_n<HB>(p.out1, n, sample0);
                        const bool ok0 = cok && zs < p.s2out.D0 && xs < p.s2out.W0;
#pragma unroll
                        for (int r = 0; r < TY; ++r) {
                            const int ys = 2 * (y0 + r) + ry;
                            const unsigned off = ((unsigned)((zs * p.s2out.H0 + ys) * p.s2out.W0 + xs) * (unsigned)(p.s2out.cin) + (unsigned)(c0)) * (unsigned)(HbEl<HB>::ES);
                            da_buf_storeq<HB>(r0, (ok0 && y0 + r < p.H && ys < p.s2out.H0) ? off : 0xFFFFFFFFu, acc[r][nn]);
                        }
                    }
                }
                if constexpr (!scattered) {
                    const long long sample = (long long)p.D * p.H * p.W * Cd;
                    const __amdgpu_buffer_rsrc_t ro = da_rsrc_n<HB>(dbase, n, sample);
#pragma unroll
                    for (int r = 0; r < TY; ++r) {
                        const unsigned off = ((unsigned)((z * p.H + (y0 + r)) * p.W + x) * (unsigned)Cd + (unsigned)cd) * HbEl<HB>::ES;
                        da_buf_storeq<HB>(ro, (cok && y0 + r < p.H) ? off : 0xFFFFFFFFu, acc[r][nn]);
                    }
                }
            }
            if (do_ep) {
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                    for (int r = 0; r < TY; ++r) acc[r][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            } else if (last) {                                       // ablated epilogue: still reset
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                    for (int r = 0; r < TY; ++r) acc[r][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (STATS && last && ((++tiles_done) & 1) == 0) stats_flush();
        if constexpr (PRO && !PRO_IN) load_pro(has_next ? 1 : 0);      // outside the branch: no vector memory in branches
        if constexpr (DYN) { if (!has_next) return false; }
        if (has_next && !(p.ablate & 4)) {
            if (p.prio_ranks == -1) __builtin_amdgcn_s_setprio(0);       // DA_PHASE_PRIO: the staging phase yields to the co-resident workgroup's K loop
            if constexpr (SP) {                    // the next tile's largest magnitude (after its prologue), one value per wave
                if constexpr (PRO && !PRO_IN) stage_pro_apply<0, PRE>(pre, vm, psc, psf, pslope);
                sp_publish(PH == 1 ? pre2 : pre);
            }
            __syncthreads();                       // every wave is done reading this item's LDS tile
            if constexpr (SP) {
                const float sn = sp_scale(nCh);
                stage_write<CK, HZ, 0, PRE, BF, SP>(lds, PH == 1 ? pre2 : pre, sn);
            }
            else if constexpr (PRO && !PRO_IN) stage_write_pro<CK, HZ, 0, PRE, BF>(lds, pre, vm, psc, psf, pslope);
            else stage_write<CK, HZ, 0, PRE, BF, SP, 0, RAW>(lds, pre);     // (PRO_IN: already transformed inside the K loop)
            if constexpr (!PRO) stage_rest(1);
            __syncthreads();
            if (p.prio_ranks == -1) __builtin_amdgcn_s_setprio(2);
        }
        cK = nK; cCh = nCh; cN = nN; cZ = nZ; cY = nY; cX = nX;
        if constexpr (SP) Ecur = Enext;
        advance();
        return true;
    };
    if constexpr (PAIR) {
#pragma unroll 1
        for (int item = 0; item < nitems; item += 2) {
            if (!item_body(item, IntC<1>{})) break;
            if (!item_body(item + 1, IntC<2>{})) break;
        }
    } else {
#pragma unroll 1
        for (int item = 0; item < nitems; ++item)
            if (!item_body(item, IntC<0>{})) break;
    }
    if (p.clk && threadIdx.x == 0) {      // DA_CLK: block 17's cycles / wall time, and the span of block lifetimes over the whole grid
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 17 && blockIdx.y == 0) { p.clk[0] = __builtin_readcyclecounter() - clk0; p.clk[1] = rt1 - rt0; }
        atomicMin(p.clk + 2, rt0); atomicMax(p.clk + 3, rt0); atomicMin(p.clk + 4, rt1); atomicMax(p.clk + 5, rt1);
        if (blockIdx.y == 0 && blockIdx.x < 1024) {      // per-workgroup record: XCC, hardware id, lifetime
            unsigned xcc, hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            p.clk[8 + 2 * blockIdx.x] = ((unsigned long long)xcc << 32) | hwid;
            p.clk[9 + 2 * blockIdx.x] = rt1 - rt0;
        }
    }
    if constexpr (STATS) {
        stats_flush();
        if ((int)threadIdx.x < NREP * 16) {
            const int co = nt0 * 16 + (int)threadIdx.x;
            if (co < p.Cout) {
                p.stats_partial[((size_t)blockIdx.x * 2 + 0) * p.Cout + co] = dsum1;
                p.stats_partial[((size_t)blockIdx.x * 2 + 1) * p.Cout + co] = dsum2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// "Thin" 3x3x3 convolutions on the VALU from an LDS halo tile: very few output channels (the 24 -> 3 flow conv,
// voxel_morph.py:57) or very few input channels (its data gradient, 3 -> 24).  An MFMA tile would be >= 13/16 padding; the
// plain direct kernel is bound by per-lane strided global gathers.  Here the 6x10x18 halo tile is staged once per channel
// chunk exactly like the MFMA kernel (buffer loads, zero padding in hardware), every thread owns two output voxels and CT
// outputs, inputs come from LDS and the weights through wave-uniform (scalar) loads.
// ---------------------------------------------------------------------------------------------------
struct ThinP {
    const float* in1; const float* in2; int C1, C2;      // C1 % CL == 0 when C2 > 0
    const float* w;                                      // zero-padded [27][CinP][CT] (thin_pack_kernel), read through the scalar cache
    const float* bias; float* out1; float* out2; int Cs1, Cs2;
    int N, D, H, W, Cout, ntz, nty, ntx, ntiles, flip_tr; float slope;
    int in_bf, out_bf;                                   // bf16 activation storage (common.h): in1 / in2, out1 / out2 are bf16 tensors (runtime: the kernel is VALU-bound)
};

// weights [27][Cin][Cout] (flip_tr: original [27][Cout][Cin], taps flipped) -> zero-padded [27][CinP][CT]: the thin kernel's inner loops
// then carry no channel predicates, and every weight address is wave-uniform
// (j0, Cw: this launch covers output channels [j0, j0 + Cout) of a layer with Cw output channels)
__global__ void thin_pack_kernel(const float* __restrict__ w, float* __restrict__ wq, int Cin, int CinP, int Cout, int CT, int flip_tr, int j0, int Cw) {
    const int total = 27 * CinP * CT;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int j = idx % CT; const int r = idx / CT; const int ci = r % CinP; const int tap = r / CinP;
        float v = 0.f;
        if (ci < Cin && j < Cout)
            v = flip_tr ? w[((size_t)(26 - tap) * Cw + j0 + j) * Cin + ci] : w[((size_t)tap * Cin + ci) * Cw + j0 + j];
        wq[idx] = v;
    }
}

// JR / CR: outputs / chunk channels that really exist (<= CT / CL): the padded ones carry zero weights, their multiply-adds are skipped
// (the 24 -> 3 flow conv and its 3 -> 24 data gradient: a quarter of the VALU work of the padded form)
template <int CL, int CT, int VPT, int JR = CT, int CR = CL>
__global__ void __launch_bounds__(256) conv3_thin_kernel(ThinP p) {
    constexpr bool WSCAL = CT <= 16;      // weights through the scalar cache (few outputs per thread) or staged in LDS (CT >= 24: too many SGPR loads)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TZ = 2 * VPT, HZ = TZ + 2, HVOX = HZ * HY * HX;
    int t = da_xcd_item_of_block((int)blockIdx.x, (int)gridDim.x);      // consecutive tiles on the same XCD: their shared halo lines are fetched into ONE L2 (x-fastest order on
                                                                        // round-robin XCDs: the first encoder's 39 MB input cost 163 MB of fabric reads, profiles/r06_step_traffic_seg.txt)
    const int tx = t % p.ntx; t /= p.ntx;
    const int ty = t % p.nty; t /= p.nty;
    const int tz = t % p.ntz; const int n = t / p.ntz;
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
    const int Cin = p.C1 + p.C2;
    const int nchunks = (Cin + CL - 1) / CL;
    const int CinP = nchunks * CL;
    // weights, zero padded to [27][CinP][CT] so the inner loops carry no channel predicates.  WSCAL: wave-uniform addresses into the padded
    // global copy -> s_load_dwordx4 through the scalar cache and SGPR operands in the FMAs, no LDS traffic (with the LDS copy 8 of
    // every 10 ds_reads of the 24 -> 3 flow conv were weights: 0.54 -> 0.45 ms).  With many outputs per thread (CT >= 24, the 3 -> 24 data
    // gradient) the scalar loads themselves become the bottleneck (0.50 -> 0.66 ms), so those keep a copy in LDS.
    const float* __restrict__ wl = p.w;
    if constexpr (!WSCAL) {
        float* wlds = lds + HVOX * CL;
        for (int idx = threadIdx.x; idx < 27 * CinP * CT; idx += 256) wlds[idx] = p.w[idx];
        wl = wlds;
    }
    // this thread's output voxels: (lz + 2 v, ly, lx), v < VPT
    const int lx = threadIdx.x & 15, ly = (threadIdx.x >> 4) & 7, lz = threadIdx.x >> 7;
    float acc[VPT][CT];
#pragma unroll
    for (int v = 0; v < VPT; ++v)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[v][j] = (p.bias && j < p.Cout) ? p.bias[j] : 0.f;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cbase = ch * CL;
        const float* src; int Cs, choff;
        if (cbase < p.C1) { src = p.in1; Cs = p.C1; choff = cbase; } else { src = p.in2; Cs = p.C2; choff = cbase - p.C1; }
        const int cvalid = (Cs - choff) < CL ? (Cs - choff) : CL;           // real channels in this chunk
        const long long sample = (long long)p.D * p.H * p.W * Cs;
        const unsigned ES = p.in_bf ? 2u : 4u;
        const __amdgpu_buffer_rsrc_t rs = p.in_bf ? da_rsrc_n<true>(src, n, sample) : da_rsrc_n<false>(src, n, sample);
        __syncthreads();
        if (CL % 4 == 0 && cvalid == CL && Cs % 4 == 0) {                   // 16-byte staging
            constexpr int Q = CL / 4 > 0 ? CL / 4 : 1;
            for (int idx = threadIdx.x; idx < HVOX * Q; idx += 256) {
                const int q = idx % Q; const int hv = idx / Q;
                const int hx = hv % HX; const int tt = hv / HX;
                const int hy = tt % HY; const int hz = tt / HY;
                const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
                const bool inb = (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                const unsigned off = ((unsigned)((z * p.H + y) * p.W + x) * (unsigned)(Cs) + (unsigned)(choff + q * 4)) * (unsigned)(ES);
                *reinterpret_cast<float4*>(lds + idx * 4) = p.in_bf ? da_buf_loadq<true>(rs, inb ? off : 0xFFFFFFFFu) : da_buf_loadq<false>(rs, inb ? off : 0xFFFFFFFFu);
            }
        } else {
            for (int idx = threadIdx.x; idx < HVOX * CL; idx += 256) {
                const int c = idx % CL; const int hv = idx / CL;
                const int hx = hv % HX; const int tt = hv / HX;
                const int hy = tt % HY; const int hz = tt / HY;
                const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
                const bool inb = c < cvalid && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                const unsigned off = ((unsigned)((z * p.H + y) * p.W + x) * (unsigned)(Cs) + (unsigned)(choff + c)) * (unsigned)(ES);
                lds[idx] = p.in_bf ? __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, inb ? off : 0xFFFFFFFFu, 0, 0) << 16)
                                   : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, inb ? off : 0xFFFFFFFFu, 0, 0));
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
            for (int dyx = 0; dyx < 9; ++dyx) {
                const int dy = dyx / 3, dx = dyx % 3;
                const float* a0 = lds + (((lz + dz) * HY + (ly + dy)) * HX + (lx + dx)) * CL;
                const float* wt = wl + ((dz * 9 + dyx) * CinP + cbase) * CT;
                float xs[VPT][CL];
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    const float* av = a0 + v * 2 * HY * HX * CL;
                    if constexpr (CL % 4 == 0) {
#pragma unroll
                        for (int c = 0; c < CL; c += 4) {
                            const float4 u = *reinterpret_cast<const float4*>(av + c);
                            xs[v][c] = u.x; xs[v][c + 1] = u.y; xs[v][c + 2] = u.z; xs[v][c + 3] = u.w;
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < CL; ++c) xs[v][c] = av[c];
                    }
                }
#pragma unroll
                for (int c = 0; c < CR; ++c) {
#pragma unroll
                    for (int j = 0; j < CT; j += 4) {
                        const float4 wv = *reinterpret_cast<const float4*>(wt + c * CT + j);
#pragma unroll
                        for (int v = 0; v < VPT; ++v) {
                            if (j < JR) acc[v][j] += xs[v][c] * wv.x;
                            if (j + 1 < JR) acc[v][j + 1] += xs[v][c] * wv.y;
                            if (j + 2 < JR) acc[v][j + 2] += xs[v][c] * wv.z;
                            if (j + 3 < JR) acc[v][j + 3] += xs[v][c] * wv.w;
                        }
                    }
                }
            }
        }
    }
    const bool vec = (p.Cs1 % 4 == 0) && (p.Cs2 % 4 == 0) && (p.Cout % 4 == 0);
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int z = z0 + lz + 2 * v, y = y0 + ly, x = x0 + lx;
        if (z >= p.D || y >= p.H || x >= p.W) continue;
        const long long vox = (((long long)n * p.D + z) * p.H + y) * p.W + x;
        if (vec) {
#pragma unroll
            for (int j = 0; j < CT; j += 4) {
                if (j < p.Cout) {
                    const float4 o = make_float4(da_act(acc[v][j], p.slope), da_act(acc[v][j + 1], p.slope),
                                                 da_act(acc[v][j + 2], p.slope), da_act(acc[v][j + 3], p.slope));
                    float* base = j < p.Cs1 ? p.out1 : p.out2;
                    const long long e = j < p.Cs1 ? vox * p.Cs1 + j : vox * p.Cs2 + (j - p.Cs1);
                    if (p.out_bf) da_stq(reinterpret_cast<da_bf16*>(base), e >> 2, o); else *reinterpret_cast<float4*>(base + e) = o;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                if (j < p.Cout) {
                    const float o = da_act(acc[v][j], p.slope);
                    float* base = j < p.Cs1 ? p.out1 : p.out2;
                    const long long e = j < p.Cs1 ? vox * p.Cs1 + j : vox * p.Cs2 + (j - p.Cs1);
                    if (p.out_bf) da_st1(reinterpret_cast<da_bf16*>(base), e, o); else base[e] = o;
                }
            }
        }
    }
}

// packed B operand: wp[chunk][step][ntile][lane][m]
__global__ void pack_fwd_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout,
                                        int CK, int NSTEPS, int NTpad, int flipped, long long total, int bf, int* zero_ctr, int nctr,
                                        int4* __restrict__ tiles, int ntiles, int ntx, int nty, int ntz, int cout0, int CoutW) {
    // (cout0, CoutW: this launch covers output channels [cout0, cout0 + Cout) of a weight tensor with CoutW output channels)
    if (zero_ctr && blockIdx.x == 0 && (int)threadIdx.x < nctr) zero_ctr[threadIdx.x] = 0;      // DYN tile counters of the launch that follows
    // tile table of the launch that follows: brick-order position -> (sample, z0, y0, x0)
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < ntiles; pos += gridDim.x * blockDim.x) {
        int n, tx, ty, tz;
        brick_tile<2, 4, 8>(pos, ntx, nty, ntz, n, tx, ty, tz);                                 // brick = 32^3 voxels
        tiles[pos] = make_int4(n, tz * 4, ty * TY, tx * TX);
    }
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(idx & 3); const int lane = (int)((idx >> 2) & 63);
        long long rest = idx >> 8;
        const int nt = (int)(rest % NTpad); rest /= NTpad;
        const int s = (int)(rest % NSTEPS); const int ch = (int)(rest / NSTEPS);
        const int g = lane >> 4, j = lane & 15;
        int tap, c4;
        if (CK == 16) { tap = s; c4 = g; } else { const int qd = s * 4 + g; tap = qd >> 1; c4 = qd & 1; }
        const int cin = ch * CK + c4 * 4 + m, cout = nt * 16 + j;
        if (bf >= 2) {      // v_mfma_f32_16x16x32_bf16: 8 bf16 per lane, lane group g = tap 2s + (g>>1), cin half g&1 (CK 16) | tap 4s + g (CK 8)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * m + h;
                const int tp = (CK == 16) ? 2 * s + (g >> 1) : 4 * s + g;
                const int ci = ch * CK + ((CK == 16) ? (g & 1) * 8 : 0) + e;
                float v2 = 0.f;
                if (tp < 27 && cout < Cout && ci < Cin)
                    v2 = flipped ? w[((size_t)(26 - tp) * CoutW + cout0 + cout) * Cin + ci] : w[((size_t)tp * Cin + ci) * CoutW + cout0 + cout];
                reinterpret_cast<unsigned short*>(wp)[idx * 2 + h] = __builtin_bit_cast(unsigned short, (__bf16)v2);
            }
            continue;
        }
        float v = 0.f;
        if (tap < 27 && cout < Cout && cin < Cin)
            v = flipped ? w[((size_t)(26 - tap) * CoutW + cout0 + cout) * Cin + cin] : w[((size_t)tap * Cin + cin) * CoutW + cout0 + cout];
        if (bf) reinterpret_cast<unsigned short*>(wp)[idx] = __builtin_bit_cast(unsigned short, (__bf16)v);      // same [..][lane][m] order, 2 bytes each
        else wp[idx] = v;
    }
}

// Split mode: the packed B operand of one 8-channel chunk -- two fp16 planes (h, l of da_split2) of 1 KiB per (K-step, N-tile), at the chunk's own
// power-of-two scale (largest |w| of the chunk in [2^14, 2^15); the exponent goes to wexp[chunk] for the kernel's accumulator bookkeeping).
// Grid (chunks, PY): every workgroup of a chunk finds the chunk's maximum (27 x 8 x Cout values, L2-resident) and packs its share of the
// (step, N-tile, lane) units with one 16-byte store per plane.  Also writes the tile table of the launch that follows (pack_fwd_weights_kernel).
struct PackJob { const float* w; unsigned short* wp; int* wexp; int4* tiles; int Cin, Cout, NTpad, flipped, ntiles, ntx, nty, ntz, cout0, CoutW; };
constexpr int kPackJobsMax = 48;
struct PackJobs { PackJob j[kPackJobsMax]; };

__device__ __forceinline__ void pack_split_weights_body(const float* __restrict__ w, unsigned short* __restrict__ wp, int* __restrict__ wexp, int Cin, int Cout,
                                                        int NTpad, int flipped, int4* __restrict__ tiles, int ntiles, int ntx, int nty, int ntz, int cout0, int CoutW) {
    __shared__ float wmax[4];
    const int nb = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
    for (int pos = bid * 256 + threadIdx.x; pos < ntiles; pos += nb * 256) {
        int n, tx, ty, tz;
        brick_tile<2, 4, 8>(pos, ntx, nty, ntz, n, tx, ty, tz);                                 // brick = 32^3 voxels
        tiles[pos] = make_int4(n, tz * 4, ty * TY, tx * TX);
    }
    const int ch = blockIdx.x;
    auto wat = [&](int tp, int ci, int cout) -> float {
        if (tp >= 27 || cout >= Cout || ci >= Cin) return 0.f;
        return flipped ? w[((size_t)(26 - tp) * CoutW + cout0 + cout) * Cin + ci] : w[((size_t)tp * Cin + ci) * CoutW + cout0 + cout];
    };
    // the chunk's largest magnitude: 16-byte loads, four in flight per thread (the weights are cold in L2 at the first call of a step)
    float m = 0.f;
    const bool vec = (Cout % 4 == 0) && (CoutW % 4 == 0) && (cout0 % 4 == 0) && (Cin % 8 == 0) && ((reinterpret_cast<size_t>(w) & 15) == 0);
    if (vec && flipped) {                 // runs of 8 consecutive ci per (tap, cout)
        const int nq = 27 * Cout * 2;
#pragma unroll 4
        for (int q = threadIdx.x; q < nq; q += 256) {
            const int r = q >> 1, tp = r / Cout, co = r - tp * Cout;
            m = da_absmax4(m, *reinterpret_cast<const float4*>(w + ((size_t)tp * CoutW + cout0 + co) * Cin + ch * 8 + (q & 1) * 4));
        }
    } else if (vec) {                     // runs of Cout consecutive couts per (tap, ci)
        const int qc = Cout / 4, nq = 27 * 8 * qc;
#pragma unroll 4
        for (int q = threadIdx.x; q < nq; q += 256) {
            const int r = q / qc, c4 = q - r * qc;
            m = da_absmax4(m, *reinterpret_cast<const float4*>(w + ((size_t)(r >> 3) * Cin + ch * 8 + (r & 7)) * CoutW + cout0 + c4 * 4));
        }
    } else if (flipped) { for (int idx = threadIdx.x; idx < 27 * Cout * 8; idx += 256) { const int e = idx & 7, r = idx >> 3; m = fmaxf(m, fabsf(wat(r / Cout, ch * 8 + e, r % Cout))); } }
    else { for (int idx = threadIdx.x; idx < 27 * 8 * Cout; idx += 256) { const int co = idx % Cout, r = idx / Cout; m = fmaxf(m, fabsf(wat(r >> 3, ch * 8 + (r & 7), co))); } }
    m = da_wave_max_nonneg(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    const int ew = da_scale_exp(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
    if (blockIdx.y == 0 && threadIdx.x == 0) wexp[ch] = ew;
    const float sc = da_pow2(ew);
    const int units = 7 * NTpad * 64;
    for (int u = blockIdx.y * 256 + threadIdx.x; u < units; u += gridDim.y * 256) {
        const int lane = u & 63, nt = (u >> 6) % NTpad, st = (u >> 6) / NTpad;
        const int tp = 4 * st + (lane >> 4), cout = nt * 16 + (lane & 15);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = wat(tp, ch * 8 + e, cout);
        uint2 h0, l0, h1, l1;
        da_split2(make_float4(v[0], v[1], v[2], v[3]), sc, h0, l0);
        da_split2(make_float4(v[4], v[5], v[6], v[7]), sc, h1, l1);
        uint4* o = reinterpret_cast<uint4*>(wp + ((size_t)((ch * 7 + st) * NTpad + nt) * 2) * 512) + lane;
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y); o[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}
__global__ void __launch_bounds__(256) pack_split_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int* __restrict__ wexp, int Cin, int Cout,
                                                                 int NTpad, int flipped, int4* __restrict__ tiles, int ntiles, int ntx, int nty, int ntz, int cout0, int CoutW) {
    pack_split_weights_body(w, wp, wexp, Cin, Cout, NTpad, flipped, tiles, ntiles, ntx, nty, ntz, cout0, CoutW);
}
// the same for many layers at once (blockIdx.z = job; grid.x = the largest chunk count): the kept packs of a whole network after an optimiser step
__global__ void __launch_bounds__(256) pack_split_weights_many_kernel(const PackJobs jobs) {
    const PackJob& j = jobs.j[blockIdx.z];
    if ((int)blockIdx.x >= j.Cin / 8) return;
    pack_split_weights_body(j.w, j.wp, j.wexp, j.Cin, j.Cout, j.NTpad, j.flipped, j.tiles, j.ntiles, j.ntx, j.nty, j.ntz, j.cout0, j.CoutW);
}

// ---------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------
struct WgP {
    const float* in1; const float* in2; int C1, C2;
    const float* dy; float* partial;
    int N, D, H, W, Cout, ntz, nty, ntx, ntiles, tiles_per_slab, O;
    unsigned masks[16]; int maskmode;   // per channel chunk tap masks (0 = all taps)
    const float* ps1; const float* pt1; const float* ps2; const float* pt2; float pslope1, pslope2;   // PRO: see FwdP
    const int4* tiles;                  // split kernel: (n, z0, y0, x0) per brick-order position (wgrad_tiles_kernel)
    int prio_ranks;                     // see da_setprio
    int ablate;                         // diagnostic only (env DA_WG_ABLATE, split kernel): 1 no staging loads, 2 no fragment reads + MFMAs, 4 no maxima / LDS writes / barriers, 16 no accumulator rescale
    S2dSrc s2in;                        // MASKED: in1 is the original tensor of a stride-2 layer, read as its space-to-depth view (cin > 0)
};

__global__ void wgrad_tiles_kernel(int4* __restrict__ tiles, int ntiles, int ntx, int nty, int ntz, int tzv) {      // tzv: z planes per tile (2 | 4)
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < ntiles; pos += gridDim.x * blockDim.x) {
        int n, tx, ty, tz;
        if (tzv == 2) brick_tile<2, 4, 16>(pos, ntx, nty, ntz, n, tx, ty, tz); else brick_tile<2, 4, 8>(pos, ntx, nty, ntz, n, tx, ty, tz);      // brick = 32^3 voxels
        tiles[pos] = make_int4(n, tz * tzv, ty * TY, tx * TX);
    }
}

// BF (bf16 matrix mode): both LDS tiles hold bf16 and one v_mfma_f32_16x16x16_bf16 consumes a whole row of 16 voxels (K = 16).
// Its fragments need 4 consecutive VOXELS of one channel per lane while the tiles are channel-contiguous, which is exactly
// what ds_read_b64_tr_b16 delivers: in each 16-lane group lane s supplies the address of (voxel 4g + s/4, channel quad s%4)
// and receives (voxels 4g .. 4g+3, channel s) -- checked in tools/ubench/ds_read_tr16.hip.
// (split mode has its own weight-gradient kernel, conv3_split_wgrad_kernel; the description that follows is the bf16 matrix mode's) K = 32 = two
// rows of 16 voxels per v_mfma_f32_16x16x32_bf16 (lane group g: row 2 rp + (g >> 1), voxels 8 (g & 1) .. + 7 = two transpose reads), six
// products per (tap slot, N-tile, row pair); the fragments of the next two tap slots are read while the current two slots' MFMAs issue.
template <int CK, int NREP, bool YS = false, bool MASKED = false, bool BF = false, bool PRO = false, bool SP = false, bool HB = false>   // YS: dY staged with dword loads (Cout % 4 != 0, e.g. the 24 -> 3 flow conv); MASKED: sparse tap sets; PRO: input prologue (BN + act applied to x while staging); HB: x and dY stored as bf16
__global__ void __launch_bounds__(256, 2) conv3_mfma_wgrad_kernel(WgP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(!HB || (BF && !SP), "bf16 activation storage: bf16 matrix mode only");
    static_assert(!(BF && YS), "bf16 mode stages dY in channel quads");
    static_assert(!SP, "split mode has its own weight-gradient kernel (conv3_split_wgrad_kernel)");
    constexpr int NP = 1;
    constexpr int TZ = 2, HZ = TZ + 2, TVOX = TZ * TY * TX;
    constexpr int CG = NREP * 16;
    constexpr int TPW = (CK == 16) ? 7 : 4;                     // tap slots per wave (CK = 8: tap PAIRS, 14 in total)
    constexpr bool SWZ = !BF && (CG % 32) == 0;
    float* ldsA = lds;
    float* ldsY = BF ? lds + HZ * HY * HX * CK / 2 * NP : lds + HZ * HY * HX * CK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4;
    const int ch = blockIdx.y, cg = blockIdx.z;
    const int cbase = ch * CK;
    const float* src; int Cs, choff;
    if (cbase < p.C1) { src = p.in1; Cs = p.C1; choff = cbase; } else { src = p.in2; Cs = p.C2; choff = cbase - p.C1; }
    unsigned vmA = 0;
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psf = make_float4(0.f, 0.f, 0.f, 0.f); float pslope = -1.f;
    if constexpr (PRO) {      // this block's channel chunk is fixed: the thread's quad constants are loaded once
        const int cofs = choff + ((int)threadIdx.x % StageGeom<CK, HZ>::Q) * 4;
        psc = *reinterpret_cast<const float4*>((cbase < p.C1 ? p.ps1 : p.ps2) + cofs);
        psf = *reinterpret_cast<const float4*>((cbase < p.C1 ? p.pt1 : p.pt2) + cofs);
        pslope = cbase < p.C1 ? p.pslope1 : p.pslope2;
    }

    // MASKED (stride 2 through space-to-depth: a chunk is one input parity and only (1|2)^3 of the 27 taps are non-zero): the live taps
    // are dealt round-robin to the waves -- slot k of wave w takes the (4k + w)-th live tap -- instead of the fixed tap = w + 4k, which
    // leaves the 8-tap parity with 3 / 3 / 1 / 1 taps per wave (and the 4-tap ones with 2 / 1 / 1 / 0).
    int slot_tap[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        if (MASKED && CK == 16) {
            unsigned m = p.masks[ch] & 0x7FFFFFFu;
            int want = 4 * k + wave, t = 27;
            while (m) { const int b = __builtin_ctz(m); m &= m - 1; if (want-- == 0) { t = b; break; } }
            slot_tap[k] = t;
        } else slot_tap[k] = wave + 4 * k;
    }
    // per-lane A offsets (floats) for this wave's tap slots
    int offA[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        int tap;
        if (CK == 16) tap = slot_tap[k]; else tap = 2 * (wave + 4 * k) + (i >> 3);
        if (BF && CK == 8) tap = 2 * (wave + 4 * k) + ((i & 3) >> 1);      // transpose-read source lane: channel quad i & 3 of the 16 (tap, ci) rows
        if (tap > 26) tap = 26;                               // garbage slot, never written out
        if (BF) offA[k] = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * CK + ((CK == 16) ? (i & 3) * 4 : (i & 1) * 4);
        else offA[k] = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * CK + ((CK == 16) ? i : (i & 7));
    }
    f32x4 acc[TPW][NREP];
#pragma unroll
    for (int k = 0; k < TPW; ++k)
#pragma unroll
        for (int nn = 0; nn < NREP; ++nn) acc[k][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // sparse tap sets (stride-2 via space-to-depth): this wave's tap slot k is live iff its tap is in the chunk's mask
    bool live[TPW];
    {
#pragma unroll
        for (int k = 0; k < TPW; ++k) live[k] = !MASKED || slot_tap[k] < 27;
    }

    const TileWalk tw = tile_walk(p.ntiles);
    const int tile_begin = 0, tile_end = tw.cnt;             // walk index k; tile = brick order position tw.lo + k * tw.J
    constexpr int NITA = StageGeom<CK, HZ>::NIT;
    constexpr int QY = CG / 4, NITY = (TVOX * QY + 255) / 256;
    float4 preA[NITA], preY[NITY];
    static_assert(!YS || NREP == 1, "scalar dY staging is only instantiated for one cout tile");
    auto tile_coords = [&](int tile, int& n, int& z0, int& y0, int& x0) {
        int tx, ty, tz;
        brick_tile<2, 4, 16>(tw.lo + tile * tw.J, p.ntx, p.nty, p.ntz, n, tx, ty, tz);               // brick = 32^3 voxels
        x0 = tx * TX; y0 = ty * TY; z0 = tz * TZ;
    };
    auto issue_loads = [&](int tile) {
        int n, z0, y0, x0;
        tile_coords(tile, n, z0, y0, x0);
        if constexpr (PRO) { vmA = 0; stage_load<CK, HZ, 0, StageGeom<CK, HZ>::NIT, HB>(preA, src, Cs, choff, n, z0, y0, x0, p.D, p.H, p.W, &vmA); }
        else if (MASKED && p.s2in.cin > 0) stage_load_s2d<CK, HZ, 0, StageGeom<CK, HZ>::NIT, HB>(preA, p.in1, p.s2in, cbase, n, z0, y0, x0, p.D, p.H, p.W);
        else stage_load<CK, HZ, 0, StageGeom<CK, HZ>::NIT, HB>(preA, src, Cs, choff, n, z0, y0, x0, p.D, p.H, p.W);
        const long long sampleY = (long long)p.D * p.H * p.W * p.Cout;
        const __amdgpu_buffer_rsrc_t ry = da_rsrc_n<HB>(p.dy, n, sampleY);
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
            int idx = threadIdx.x + it * 256;
            asm volatile("" : "+v"(idx));
            const int c4 = idx % QY; const int v = idx / QY;
            const int x = x0 + (v & 15), y = y0 + ((v >> 4) & 7), z = z0 + (v >> 7);
            const int co = cg * CG + c4 * 4;
            const bool vin = idx < TVOX * QY && z < p.D && y < p.H && x < p.W;
            const unsigned off = ((unsigned)((z * p.H + y) * p.W + x) * (unsigned)(p.Cout) + (unsigned)(co)) * (unsigned)(HbEl<HB>::ES);
            if constexpr (!YS) {
                preY[it] = da_buf_loadq<HB>(ry, (vin && co < p.Cout) ? off : 0xFFFFFFFFu);
            } else {
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    t[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, (vin && co + j < p.Cout) ? off + 4 * j : 0xFFFFFFFFu, 0, 0));
                preY[it] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
    };
    auto write_lds = [&]() {
        if constexpr (PRO) stage_write_pro<CK, HZ, 0, StageGeom<CK, HZ>::NIT, BF>(ldsA, preA, vmA, psc, psf, pslope);
        else stage_write<CK, HZ, 0, StageGeom<CK, HZ>::NIT, BF>(ldsA, preA);
        // dY tile [TVOX][CG] (channel halves XOR-swizzled by voxel parity when CG % 32 == 0)
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int c4 = idx % QY; const int v = idx / QY;
            int c = c4 * 4;
            if (SWZ) c ^= (v & 1) << 4;
            if (idx < TVOX * QY) {
                if constexpr (BF) reinterpret_cast<uint2*>(ldsY)[idx] = make_uint2(da_bf16x2(preY[it].x, preY[it].y), da_bf16x2(preY[it].z, preY[it].w));   // linear [v][CG] in bf16
                else *reinterpret_cast<float4*>(ldsY + v * CG + c) = preY[it];
            }
        }
    };
    if (tile_begin < tile_end) { issue_loads(tile_begin); write_lds(); }
    __syncthreads();
    const int prio_rank = (int)((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) / 256u);
#pragma unroll 1
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        if (p.prio_ranks > 1) da_setprio((prio_rank + tile) % p.prio_ranks);
        const bool has_next = tile + 1 < tile_end;
        if (has_next) issue_loads(tile + 1);                 // next tile's global loads fly during this tile's MFMAs
        // K loop: 16 rows (vz, vy) x 4 K-steps (4 voxels along x each).  Per row one base address per operand; the four
        // steps use compile-time offsets (ds_read immediates), so the VALU work per 28*NREP MFMAs is a handful of adds.
        // Fragments of step j+1 are read while the MFMAs of step j issue (hipcc otherwise serialises read->wait->mfma).
        const int swz = SWZ ? ((g & 1) << 4) : 0;             // voxel parity == g & 1 (row base and 4*j are even)
        if constexpr (BF) {
            typedef s16x4 __attribute__((address_space(3))) * lds_frag_ptr;
            const short* ldsAh = reinterpret_cast<const short*>(ldsA);
            const short* ldsYh = reinterpret_cast<const short*>(ldsY);
#pragma unroll 2
            for (int row = 0; row < TVOX / 16; ++row) {
                const int vz = row >> 3, vy = row & 7;
                const short* arow = ldsAh + ((vz * HY + vy) * HX + 4 * g + (i >> 2)) * CK;
                const short* yrow = ldsYh + (row * 16 + 4 * g + (i >> 2)) * CG + (i & 3) * 4;
                s16x4 a[TPW], b[NREP];
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn) b[nn] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)(yrow + nn * 16));
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if (!MASKED || live[k]) a[k] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)(arow + offA[k]));
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if (!MASKED || live[k]) {
#pragma unroll
                        for (int nn = 0; nn < NREP; ++nn)
                            acc[k][nn] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[k], b[nn], acc[k][nn], 0, 0, 0);
                    }
            }
        } else
#pragma unroll 1
        for (int row = 0; row < TVOX / 16; ++row) {
            const int vz = row >> 3, vy = row & 7;
            const float* arow = ldsA + ((vz * HY + vy) * HX + g) * CK;
            const float* yrow = ldsY + (row * 16 + g) * CG + i;
            float a0[TPW], b0[NREP];
#pragma unroll
            for (int nn = 0; nn < NREP; ++nn) b0[nn] = yrow[(nn * 16) ^ swz];
#pragma unroll
            for (int k = 0; k < TPW; ++k) a0[k] = (!MASKED || live[k]) ? arow[offA[k]] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a1[TPW], b1[NREP];
                if (j < 3) {
#pragma unroll
                    for (int nn = 0; nn < NREP; ++nn) b1[nn] = yrow[(j + 1) * 4 * CG + ((nn * 16) ^ swz)];
#pragma unroll
                    for (int k = 0; k < TPW; ++k) a1[k] = (!MASKED || live[k]) ? arow[(j + 1) * 4 * CK + offA[k]] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < TPW; ++k)
                    if (!MASKED || live[k]) {
#pragma unroll
                        for (int nn = 0; nn < NREP; ++nn)
                            acc[k][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[k], b0[nn], acc[k][nn], 0, 0, 0);
                    }
                if (j < 3) {
#pragma unroll
                    for (int k = 0; k < TPW; ++k) a0[k] = a1[k];
#pragma unroll
                    for (int nn = 0; nn < NREP; ++nn) b0[nn] = b1[nn];
                }
            }
        }
        if (has_next) {
            __syncthreads();
            write_lds();
            __syncthreads();
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
                if (CK == 16) { tap = slot_tap[k]; ci = row; } else { tap = 2 * (wave + 4 * k) + (row >> 3); ci = row & 7; }
                if (tap < 27 && co < p.Cout && (CK == 16 || wave + 4 * k < 14))
                    part[((size_t)tap * Cin + cbase + ci) * p.Cout + co] = acc[k][nn][reg];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient in split mode, second form (the default): every wave owns two output rows of the 2 x 8 x 16 tile and ALL 27 taps.
//   * K = 32 per v_mfma_f32_16x16x32_bf16 = 16 voxels along x times the tile's two z planes (lane group g: plane g >> 1, voxels
//     8 (g & 1) .. + 7), so a shift along y moves whole fragments: the x fragment of halo row h serves (row, dy) = (h, 0), (h - 1, 1),
//     (h - 2, 2) -- four fragments per tap class instead of six;
//   * M = 16 rows = two taps x 8 cin that share dy: the nine (dz, dx) combinations form four pairs and one single (class 4, upper half
//     idle: 27 taps in 30 slots);
//   * the dY fragments of the wave's two rows are read once per tile and stay in registers for all 15 (class, dy) accumulators.
// LDS reads per MFMA fall from ~1.3 to ~0.7 ds_read_b64_tr_b16, all four waves carry the same load, and the accumulators (per-wave
// partial sums over the wave's rows) are reduced across the waves through LDS once, at the end of the persistent loop.
// ---------------------------------------------------------------------------------------------------
// NPL = 2: split mode (da_split2: two fp16 planes of x and of dY at the tile's own power-of-two scales, three products per fragment pair; the
// accumulators live in the unit 2^E of the tile being accumulated, E = x exponent + dY exponent capped at 40 above the smallest E of the
// slab so far, and are rescaled by the exact factor 2^(E - E_previous) between tiles).  NPL = 1: bf16 matrix mode -- operands rounded to
// bf16, ONE product; the same LDS geometry and accumulator layout at a third of the matrix work, i.e. bound by its staging (the older bf16
// weight-gradient kernel issues v_mfma_f32_16x16x16_bf16, half the rate of the K = 32 form).  HB: x and dY stored as bf16 (bf16 activation
// storage; NPL = 1 only).
// Quads of padding after every z plane of the x tile.  The fragment read of the tap-pair class that mixes two z planes (combos 2 and 3:
// lanes q < 2 read plane dz, lanes q >= 2 plane dz + 1) then starts 16 banks apart instead of 8 within each half wave: 48 -> 16 weight
// gradient 2.438 -> 2.395 ms, 96 -> 32 1.431 -> 1.406 (2 and 8 quads: less; 0 = the unpadded image).
#ifndef DA_WG_ZPAD
#define DA_WG_ZPAD 4
#endif
#ifndef DA_WG_TZ
#define DA_WG_TZ 2      // z planes per tile of the split weight gradient (4: two z-plane pairs per staged tile -- measured 48 -> 16 2.01 -> 1.85 ms but 16 -> 16 0.64 -> 0.70: spills at 256 VGPRs; kept for A/B builds)
#endif
// (Tried: a 1-D launch that gives the two workgroups of a CU -- blocks L and L + 256, tools/ubench/wg_placement.hip -- neighbouring chunks of
// one slab so that the one behind finds the other's lines in L1.  No effect, 2.008 vs 2.017 ms on 48 -> 16 on one box: a tile's lines, 92 KB,
// pass through the 32 KB L1 long before the partner asks for them.)
template <bool PRO, int NPL = 2, bool HB = false>
__global__ void __launch_bounds__(256, 2) conv3_split_wgrad_kernel(WgP p) {
    static_assert(!HB || NPL == 1, "bf16 activation storage goes with the bf16 matrix mode");
    static_assert(NPL == 1 || NPL == 2, "one bf16 plane or the two fp16 planes of the split mode");
    constexpr bool SPL = NPL == 2;
    using WFrag = std::conditional_t<SPL, f16x8, bf16x8>;
    constexpr bool RAWA = HB && !SPL && !PRO, RAWY = HB && !SPL;       // bf16 tensors copied straight into the bf16 LDS image (da_buf_loadq)
    constexpr unsigned ES = HbEl<HB>::ES;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CK = 8, CG = 16, TZ = DA_WG_TZ, HZ = TZ + 2, TVOX = TZ * TY * TX;      // tile 4 x 8 x 16: two z-plane pairs per staged tile (half the barriers and tile-table reads per MFMA, halo 2.1x instead of 2.8x)
    static_assert(TZ == 2 || TZ == 4, "one or two z-plane pairs per tile");
    constexpr int ZPQ = DA_WG_ZPAD, ZPE = 4 * ZPQ;                         // padding after every z plane of the x tile: quads / elements
    constexpr int PLA = HZ * (HY * HX * CK + ZPE), PLY = TVOX * CG;        // elements per plane
    float* ldsA = lds;
    float* ldsY = lds + PLA / 2 * NPL;
    typedef s16x4 __attribute__((address_space(3))) * lds_frag_ptr;
    const short* ldsAh = reinterpret_cast<const short*>(ldsA);
    const short* ldsYh = reinterpret_cast<const short*>(ldsY);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4, q = i & 3, vq = i >> 2;
    const int slab = blockIdx.x, nsl = gridDim.x, ch = blockIdx.y, cg = blockIdx.z;
    const int cbase = ch * CK;
    const float* src; int Cs, choff;
    if (cbase < p.C1) { src = p.in1; Cs = p.C1; choff = cbase; } else { src = p.in2; Cs = p.C2; choff = cbase - p.C1; }
    unsigned vmA = 0;
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psf = make_float4(0.f, 0.f, 0.f, 0.f); float pslope = -1.f;
    if constexpr (PRO) {
        const int cofs = choff + ((int)threadIdx.x % StageGeom<CK, HZ>::Q) * 4;
        psc = *reinterpret_cast<const float4*>((cbase < p.C1 ? p.ps1 : p.ps2) + cofs);
        psf = *reinterpret_cast<const float4*>((cbase < p.C1 ? p.pt1 : p.pt2) + cofs);
        pslope = cbase < p.C1 ? p.pslope1 : p.pslope2;
    }
    // transpose-read source of this lane: voxel 8 (g & 1) + vq [+ 4] of plane g >> 1, channel quad q = (tap half q >> 1, cin quad q & 1)
    const int laneA = ((((g >> 1) * HY) + 2 * wave) * HX + 8 * (g & 1) + vq) * CK + (q & 1) * 4 + (g >> 1) * ZPE;
    int offC[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const int combo = (c < 4) ? 2 * c + (q >> 1) : 8;                   // (class 4, upper tap half: re-reads tap 8; its rows are never written)
        offC[c] = ((combo / 3) * HY * HX + combo % 3) * CK + (combo / 3) * ZPE;
    }
    const int laneY = ((((g >> 1) * TY) + 2 * wave) * TX + 8 * (g & 1) + vq) * CG + q * 4;
    auto tr8 = [&](const short* a, int step) -> WFrag {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)a);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)(a + step));
        return __builtin_bit_cast(WFrag, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma = [&](f32x4 c, const WFrag& a, const WFrag& b) -> f32x4 {
        if constexpr (SPL) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    };
    struct F3 { WFrag p[NPL]; };
    int zpA = 0, zpY = 0;                                                  // element offsets of the z-plane pair being accumulated
    auto loadF = [&](int c, int h) -> F3 {                                  // x fragment of class c, halo row 2 wave + h
        F3 f; const short* a = ldsAh + laneA + zpA + offC[c] + h * (HX * CK);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) f.p[pl] = tr8(a + pl * PLA, 4 * CK);
        return f;
    };
    // class 4 (the unpaired combo 8): its idle upper tap half takes the SAME combo one halo row further down, i.e. dy + 1 -- fragment of
    // halo rows (2 wave + h | 2 wave + h + 1).  Its three dy then need two accumulators instead of three: 14 instead of 15 MFMA groups per
    // row pair (27 taps in 28 slots instead of 30).
    auto loadG = [&](int h) -> F3 {
        F3 f; const short* a = ldsAh + laneA + zpA + offC[4] + (h + (q >> 1)) * (HX * CK);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) f.p[pl] = tr8(a + pl * PLA, 4 * CK);
        return f;
    };
    auto loadY = [&](int r) -> F3 {                                         // dY fragment of output row 2 wave + r
        F3 f; const short* a = ldsYh + laneY + zpY + r * (TX * CG);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) f.p[pl] = tr8(a + pl * PLY, 4 * CG);
        return f;
    };
    f32x4 acc[5][3];
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const TileWalk tw = tile_walk(p.ntiles, nsl, slab);
    constexpr int NITA = StageGeom<CK, HZ>::NIT, QY = CG / 4, NITY = (TVOX * QY + 255) / 256;
    float4 preA[NITA], preY[NITY];
    // Per-thread constants of the two staging patterns (the tile coordinates come from the table the launcher's tile kernel wrote):
    // x halo: StageMap; dY: iteration `it` covers voxel v = it * 64 + threadIdx.x / 4, cout quad threadIdx.x % 4.
    StageMap<CK, HZ> smap; smap.init(p.H, p.W);
    const int yq4 = (cg * CG + ((int)threadIdx.x % QY) * 4);
    const int yv0 = (int)threadIdx.x / QY;
    int voY[NITY];                                            // dY voxel index relative to the tile's corner: constant for the whole launch
    const bool smallY = (long long)TZ * p.H * p.W < (1ll << 24) && (long long)p.Cout * ES < (1ll << 24);
#pragma unroll
    for (int it = 0; it < NITY; ++it) { const int v = yv0 + it * (256 / QY); voY[it] = ((v >> 7) * p.H + ((v >> 4) & 7)) * p.W + (v & 15); }
    // The tile table is read with a vector load (hipcc does not prove the table invariant): an entry is requested one tile before the
    // staging loads that need it, so its latency never sits in front of them.
    auto fetch_tile = [&](int tile) -> int4 {
        int pos = tw.lo + tile * tw.J; pos = pos < p.ntiles ? pos : p.ntiles - 1;
        return p.tiles[__builtin_amdgcn_readfirstlane(pos)];
    };
    auto issue_loads = [&](const int4 tv) {
        const int n = __builtin_amdgcn_readfirstlane(tv.x), z0 = __builtin_amdgcn_readfirstlane(tv.y), y0 = __builtin_amdgcn_readfirstlane(tv.z), x0 = __builtin_amdgcn_readfirstlane(tv.w);
        {
            const long long sample = (long long)p.D * p.H * p.W * Cs;
            const __amdgpu_buffer_rsrc_t rs = da_rsrc_n<HB>(src, n, sample);
            const typename StageMap<CK, HZ>::Tile st = smap.tile(z0, y0, x0, p.D, p.H, p.W, Cs, choff, true, (int)ES);
            if constexpr (PRO) vmA = 0;
#pragma unroll
            for (int it = 0; it < NITA; ++it) {
                const unsigned so = smap.offset(st, it);
                preA[it] = da_buf_loadq<HB, RAWA>(rs, so);
                if constexpr (PRO) vmA |= (so != 0xFFFFFFFFu ? 1u : 0u) << it;
            }
        }
        const long long sampleY = (long long)p.D * p.H * p.W * p.Cout;
        const __amdgpu_buffer_rsrc_t ry = da_rsrc_n<HB>(p.dy, n, sampleY);
        const bool inside = smallY && z0 + TZ <= p.D && y0 + TY <= p.H && x0 + TX <= p.W && cg * CG + CG <= p.Cout;      // wave-uniform: the whole dY tile exists
        const int basev = (z0 * p.H + y0) * p.W + x0;
        const unsigned baseY = ((unsigned)basev * (unsigned)p.Cout + (unsigned)yq4) * ES;
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
            const int v = yv0 + it * (256 / QY);
            const int vx = v & 15, vy = (v >> 4) & 7, vz = v >> 7;
            unsigned off;
            if (inside) off = __umul24((unsigned)voY[it], (unsigned)p.Cout * ES) + baseY;
            else {
                const int x = x0 + vx, y = y0 + vy, z = z0 + vz;
                const bool vin = z < p.D && y < p.H && x < p.W && yq4 < p.Cout;
                off = vin ? (unsigned)((((z * p.H + y) * p.W + x) * p.Cout + yq4) * ES) : 0xFFFFFFFFu;
            }
            preY[it] = da_buf_loadq<HB, RAWY>(ry, off);
        }
    };
    // SPL: scale bookkeeping (wave-uniform).  The accumulators hold (true sums) x 2^Eacc; Emin = smallest E of this slab so far.
    float* smax = ldsY + PLY / 2 * NPL;                       // [2][4]: the four waves' largest |x| and |dY| of the tile about to be written
    int Eacc = 0, Emin = 0, Enext = 0; bool first_tile = true;
    auto publish_max = [&]() {                                // before the barrier that retires the current tile
        if constexpr (SPL) {
            if constexpr (PRO) stage_pro_apply<0, NITA>(preA, vmA, psc, psf, pslope);
            const float ma = da_wave_max_nonneg(stage_absmax<NITA>(preA)), my = da_wave_max_nonneg(stage_absmax<NITY>(preY));
            if (lane == 0) { smax[wave] = ma; smax[4 + wave] = my; }
        }
    };
    auto write_lds = [&]() {
        float sa = 1.f, sy = 1.f;
        if constexpr (SPL) {
            const float4 ma = *reinterpret_cast<const float4*>(smax), my = *reinterpret_cast<const float4*>(smax + 4);
            const int ea = da_scale_exp(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fmaxf(fmaxf(ma.x, ma.y), fmaxf(ma.z, ma.w))))));
            const int ey = da_scale_exp(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fmaxf(fmaxf(my.x, my.y), fmaxf(my.z, my.w))))));
            int E = ea + ey;
            if (!first_tile) E = min(E, Emin + 40);
            // keep the accumulators' unit while this tile fits it: a tile up to 8x smaller than the unit allows is staged at the unit's scale
            // (its largest value then lies in [2^11, 2^15) instead of [2^14, 2^15): the per-product bound is unchanged, only the floor below
            // which l underflows rises from 2^-40 to 2^-37 of the tile maximum), so the 60 accumulator multiplies run only when the data's
            // magnitude really moves -- they cost 0.19 of 1.98 ms on the 48 -> 16 layer when done every tile
            if (!first_tile && E >= Eacc && E <= Eacc + 3 && !(p.ablate & 32)) E = Eacc;
            Emin = first_tile ? E : min(Emin, E);
            first_tile = false;
            Enext = E;
            sy = da_pow2(ey); sa = da_pow2(E - ey);
            stage_write<CK, HZ, 0, NITA, true, true, ZPQ>(ldsA, preA, sa);
        }
        else if constexpr (PRO) stage_write_pro<CK, HZ, 0, NITA, true, ZPQ>(ldsA, preA, vmA, psc, psf, pslope);
        else stage_write<CK, HZ, 0, NITA, true, false, ZPQ, RAWA>(ldsA, preA);
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
            const int idx = threadIdx.x + it * 256;
            if (idx < TVOX * QY) {
                if constexpr (SPL) {
                    uint2 h, l; da_split2(preY[it], sy, h, l);
                    reinterpret_cast<uint2*>(ldsY)[idx] = h; reinterpret_cast<uint2*>(ldsY)[idx + TVOX * QY] = l;
                } else if constexpr (RAWY) reinterpret_cast<uint2*>(ldsY)[idx] = make_uint2(__float_as_uint(preY[it].x), __float_as_uint(preY[it].y));
                else reinterpret_cast<uint2*>(ldsY)[idx] = make_uint2(da_bf16x2(preY[it].x, preY[it].y), da_bf16x2(preY[it].z, preY[it].w));
            }
        }
    };
    int4 tnext = fetch_tile(1);
    if (tw.cnt > 0) { issue_loads(fetch_tile(0)); publish_max(); if constexpr (SPL) __syncthreads(); write_lds(); }
    __syncthreads();
    constexpr int NPR = SPL ? 3 : 1;
    constexpr int PA[3] = {0, SPL ? 1 : 0, 0}, PB[3] = {SPL ? 1 : 0, 0, 0};      // (x, dY) plane pairs, small terms first (one plane: the single product)
    const int prio_rank = (int)((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) / 256u);
#pragma unroll 1
    for (int tile = 0; tile < tw.cnt; ++tile) {
        if (p.prio_ranks > 1) da_setprio((prio_rank + tile) % p.prio_ranks);
        const bool has_next = tile + 1 < tw.cnt;
        if (has_next && !(p.ablate & 1)) issue_loads(tnext);    // next tile's global loads fly during this tile's MFMAs
        tnext = fetch_tile(tile + 2);
        if (SPL && Enext != Eacc && !(p.ablate & 16)) {         // bring the running sums into this tile's unit (exact: a power of two)
            const float f = da_acc_factor(Enext - Eacc);
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) acc[c][d] = acc[c][d] * f;
            Eacc = Enext;
        }
        if (!(p.ablate & 2))
#pragma unroll
        for (int zp = 0; zp < TZ / 2; ++zp) {                   // the tile's z-plane pairs (K = 16 voxels x the two planes of a pair)
        zpA = zp * 2 * (HY * HX * CK + ZPE); zpY = zp * 2 * (TY * TX * CG);
        F3 Y0 = loadY(0), Y1 = loadY(1);
        F3 Fa = loadF(0, 0), Fb = loadF(0, 1), Fc = loadF(0, 2), Fd, Na, Nb, Nc;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            Fd = loadF(c, 3);
            Na = (c < 3) ? loadF(c + 1, 0) : loadG(0);
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {                  // output row 2 wave: halo rows 0, 1, 2 <-> dy 0, 1, 2
                acc[c][0] = mma(acc[c][0], Fa.p[PA[pr]], Y0.p[PB[pr]]);
                acc[c][1] = mma(acc[c][1], Fb.p[PA[pr]], Y0.p[PB[pr]]);
                acc[c][2] = mma(acc[c][2], Fc.p[PA[pr]], Y0.p[PB[pr]]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (c < 3) { Nb = loadF(c + 1, 1); Nc = loadF(c + 1, 2); } else { Nb = loadG(1); Nc = loadF(4, 2); }
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {                  // output row 2 wave + 1: halo rows 1, 2, 3
                acc[c][0] = mma(acc[c][0], Fb.p[PA[pr]], Y1.p[PB[pr]]);
                acc[c][1] = mma(acc[c][1], Fc.p[PA[pr]], Y1.p[PB[pr]]);
                acc[c][2] = mma(acc[c][2], Fd.p[PA[pr]], Y1.p[PB[pr]]);
            }
            __builtin_amdgcn_sched_barrier(0);
            Fa = Na; Fb = Nb; Fc = Nc;
        }
        {   // class 4: Fa = rows (0 | 1), Fb = rows (1 | 2), Fc = row 2, Fd = row 3 of combo 8; acc[4][0] = (dy 0 | dy 1), acc[4][1] = (dy 2 | -)
            Fd = loadF(4, 3);
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                acc[4][0] = mma(acc[4][0], Fa.p[PA[pr]], Y0.p[PB[pr]]);
                acc[4][1] = mma(acc[4][1], Fc.p[PA[pr]], Y0.p[PB[pr]]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                acc[4][0] = mma(acc[4][0], Fb.p[PA[pr]], Y1.p[PB[pr]]);
                acc[4][1] = mma(acc[4][1], Fd.p[PA[pr]], Y1.p[PB[pr]]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (has_next && !(p.ablate & 4)) {
            if (p.prio_ranks == -1) __builtin_amdgcn_s_setprio(0);
            publish_max();
            __syncthreads();
            write_lds();
            __syncthreads();
            if (p.prio_ranks == -1) __builtin_amdgcn_s_setprio(2);
        }
    }
    if constexpr (SPL) {                                        // back to the true unit (two exact factors: |E| may exceed 127)
        const float inv1 = da_pow2(-(Eacc / 2)), inv2 = da_pow2(-(Eacc - Eacc / 2));
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) acc[c][d] = acc[c][d] * inv1 * inv2;
    }
    // reduce the four waves' partial sums through LDS (two rounds of <= 30 KB), then wave 0 writes this slab's partial dW
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(lds);
    auto put = [&](int slot) {
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) red[((slot * 15) + c * 3 + d) * 64 + lane] = make_float4(acc[c][d][0], acc[c][d][1], acc[c][d][2], acc[c][d][3]);
    };
    auto add = [&](int slot) {
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float4 v = red[((slot * 15) + c * 3 + d) * 64 + lane];
                acc[c][d][0] += v.x; acc[c][d][1] += v.y; acc[c][d][2] += v.z; acc[c][d][3] += v.w;
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
        float* part = p.partial + (size_t)slab * p.O;
        const int Cin = p.C1 + p.C2;
        const int co = cg * CG + i;
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 4 * g + reg;
                    // classes 0 - 3: (combo 2c | 2c + 1) at dy = d; class 4: combo 8 at dy = (0 | 1) in accumulator 0, dy = (2 | -) in accumulator 1
                    const int combo = c < 4 ? 2 * c + (row >> 3) : 8;
                    const int dyt = c < 4 ? d : 2 * d + (row >> 3);
                    const int tap = (combo / 3) * 9 + dyt * 3 + combo % 3, ci = row & 7;
                    if (dyt < 3 && (c < 4 || d < 2) && co < p.Cout) part[((size_t)tap * Cin + cbase + ci) * p.Cout + co] = acc[c][d][reg];
                }
    }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient in split mode, third form: 16-channel chunks, eight waves.
// The staging loads of the row-owner kernel above fetch 32 bytes (8 channels) of every voxel while the L1 asks the L2 for 64-byte
// sectors, and the kernel is bound by exactly that stream: with everything but the loads removed it runs 1.91 of 1.98 ms (48 -> 16,
// DA_WG_ABLATE=6); the counters show 167 M sector requests per launch at 387 cycles of latency and the L1 stalled on its outstanding
// requests 35 % of the time -- ~64 sectors in flight x 64 B / 387 cycles = the observed ~10 B / clk / CU (profiles/r04_wgrad_memory_path.txt).
// Here a workgroup stages 16 channels = one whole sector per voxel (half the requests for the same data, and half as many passes over
// the dY tile), as EIGHT waves: waves 0 - 3 own the chunk's first 8 channels, waves 4 - 7 the second 8, each exactly the row-owner
// kernel's per-wave program (two output rows, all 27 taps, 14 accumulators) on the shared dY tile -- so the register budget per wave is
// unchanged, and per thread the staging shrinks (6 x quads as before, 2 dY quads instead of 4).  One workgroup per CU (62 KB of LDS,
// 2 waves per SIMD as before); chosen when C1 and C2 are multiples of 16 and slabs x chunks x cout groups fills >= 224 of the 256 CUs.
// ---------------------------------------------------------------------------------------------------
#ifndef DA_WG16_ZPAD
#define DA_WG16_ZPAD 4
#endif
// Phases as in the row-owner kernel (MFMAs | barrier | convert + write | barrier), ~205 registers and 62 KB of LDS.  A two-buffer form (one
// barrier per tile, the next tile converted between the MFMAs, loads two tiles ahead: 254 registers, 125 KB) is faster alone (48 -> 16:
// 1.72 vs 1.85 ms) but slower in the training step (seg 22.8 vs 21.65 ms): it fills the register file, and the BatchNorm-backward kernels
// of the main stream -- HBM-bound, the natural partners of a matrix-bound weight gradient on the side stream -- then wait for it to end
// instead of running beside it in the 2 x 48 registers per SIMD and 98 KB of LDS this form leaves free.
// NPL = 2: split mode (fp32 tensors, two fp16 planes, three products).  NPL = 1 with HB: bf16 activation storage in the bf16 matrix mode -- one
// bf16 plane, one product, no scales; a 16-channel chunk is then 32 bytes of a voxel (the row-owner form stages 16: a quarter of a sector).
template <bool PRO, int NPL = 2, bool HB = false>
__global__ void __launch_bounds__(512, 1) conv3_split_wgrad16_kernel(WgP p) {
    static_assert(NPL == 2 || (NPL == 1 && HB), "two fp16 planes of fp32 tensors, or one bf16 plane of bf16 tensors");
    constexpr bool SPL = NPL == 2;
    using WFrag = std::conditional_t<SPL, f16x8, bf16x8>;
    constexpr bool RAWA = HB && !SPL && !PRO, RAWY = HB && !SPL;       // bf16 tensors copied straight into the bf16 LDS image
    constexpr unsigned ES = HbEl<HB>::ES;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CK = 16, CG = 16, TZ = 2, HZ = TZ + 2, TVOX = TZ * TY * TX, NT = 512;
    constexpr int ZPQ = DA_WG16_ZPAD, ZPE = 4 * ZPQ;
    constexpr int QA = CK / 4, HV = HZ * HY * HX, TOTA = HV * QA, NITA = (TOTA + NT - 1) / NT;
    constexpr int PLH = HZ * (HY * HX * 8 + ZPE), PLA = 2 * PLH, PLY = TVOX * CG;      // elements per half image / per plane (x tile: the chunk's two 8-channel halves as two images of 16-byte voxel records -- the row-owner kernel's layout, 1.2 LDS cycles per half-wave fragment read; 32-byte records with the halves side by side: 2.8, lanes 8 voxels apart are then exactly 64 banks apart)
    constexpr int QY = CG / 4, NITY = (TVOX * QY + NT - 1) / NT;
    float* ldsA = lds;
    float* ldsY = lds + PLA / 2 * NPL;                                     // NPL planes of PLA two-byte elements
    float* smax = ldsY + PLY / 2 * NPL;                                    // [2][8]
    typedef s16x4 __attribute__((address_space(3))) * lds_frag_ptr;
    const short* ldsAh = reinterpret_cast<const short*>(ldsA);
    const short* ldsYh = reinterpret_cast<const short*>(ldsY);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wave & 3, wh = wave >> 2;                               // row pair, channel half
    const int i = lane & 15, g = lane >> 4, q = i & 3, vq = i >> 2;
    const int slab = blockIdx.x, nsl = gridDim.x, ch = blockIdx.y, cg = blockIdx.z;
    const int cbase = ch * CK;
    const float* src; int Cs, choff;
    if (cbase < p.C1) { src = p.in1; Cs = p.C1; choff = cbase; } else { src = p.in2; Cs = p.C2; choff = cbase - p.C1; }
    const int c4 = (int)threadIdx.x % QA;
    unsigned vmA = 0;
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psf = make_float4(0.f, 0.f, 0.f, 0.f); float pslope = -1.f;
    if constexpr (PRO) {
        const int cofs = choff + c4 * 4;
        psc = *reinterpret_cast<const float4*>((cbase < p.C1 ? p.ps1 : p.ps2) + cofs);
        psf = *reinterpret_cast<const float4*>((cbase < p.C1 ? p.pt1 : p.pt2) + cofs);
        pslope = cbase < p.C1 ? p.pslope1 : p.pslope2;
    }
    // fragment sources exactly as in conv3_split_wgrad_kernel, in this wave's half image
    const int laneA = ((((g >> 1) * HY) + 2 * wr) * HX + 8 * (g & 1) + vq) * 8 + (q & 1) * 4 + (g >> 1) * ZPE + wh * PLH;
    int offC[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const int combo = (c < 4) ? 2 * c + (q >> 1) : 8;
        offC[c] = ((combo / 3) * HY * HX + combo % 3) * 8 + (combo / 3) * ZPE;
    }
    const int laneY = ((((g >> 1) * TY) + 2 * wr) * TX + 8 * (g & 1) + vq) * CG + q * 4;
    auto tr8 = [&](const short* a, int step) -> WFrag {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)a);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)(a + step));
        return __builtin_bit_cast(WFrag, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma = [&](f32x4 c, const WFrag& a, const WFrag& b) -> f32x4 {
        if constexpr (SPL) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    };
    struct F3 { WFrag p[NPL]; };
    auto loadF = [&](int c, int h) -> F3 {
        F3 f; const short* a = ldsAh + laneA + offC[c] + h * (HX * 8);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) f.p[pl] = tr8(a + pl * PLA, 4 * 8);
        return f;
    };
    auto loadG = [&](int h) -> F3 {
        F3 f; const short* a = ldsAh + laneA + offC[4] + (h + (q >> 1)) * (HX * 8);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) f.p[pl] = tr8(a + pl * PLA, 4 * 8);
        return f;
    };
    auto loadY = [&](int r) -> F3 {
        F3 f; const short* a = ldsYh + laneY + r * (TX * CG);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) f.p[pl] = tr8(a + pl * PLY, 4 * CG);
        return f;
    };
    f32x4 acc[5][3];
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const TileWalk tw = tile_walk(p.ntiles, nsl, slab);
    float4 preA[NITA], preY[NITY];
    // staging maps (launch constants): x halo quad idx = threadIdx.x + 512 it -> halo voxel idx / 4; dY quad idx -> voxel idx / 4
    int voA[NITA]; unsigned pkA[NITA];
#pragma unroll
    for (int it = 0; it < NITA; ++it) {
        const int hv = ((int)threadIdx.x + it * NT) / QA;
        const int hx = hv % HX, t = hv / HX, hy = t % HY, hz = t / HY;
        pkA[it] = (hv < HV) ? ((unsigned)hz << 16 | (unsigned)hy << 8 | (unsigned)hx) : 0xFFFF0000u;
        voA[it] = (hv < HV) ? (hz * p.H + hy) * p.W + hx : 0;
    }
    const bool smallA = (long long)HZ * p.H * p.W < (1ll << 24) && (long long)Cs * ES < (1ll << 24);
    const int yq4 = cg * CG + ((int)threadIdx.x % QY) * 4;
    const int yv0 = (int)threadIdx.x / QY;
    int voY[NITY];
    const bool smallY = (long long)TZ * p.H * p.W < (1ll << 24) && (long long)p.Cout * ES < (1ll << 24);
#pragma unroll
    for (int it = 0; it < NITY; ++it) { const int v = yv0 + it * (NT / QY); voY[it] = ((v >> 7) * p.H + ((v >> 4) & 7)) * p.W + (v & 15); }
    auto fetch_tile = [&](int tile) -> int4 {
        int pos = tw.lo + tile * tw.J; pos = pos < p.ntiles ? pos : p.ntiles - 1;
        return p.tiles[__builtin_amdgcn_readfirstlane(pos)];
    };
    auto issue_loads = [&](const int4 tv) {
        const int n = __builtin_amdgcn_readfirstlane(tv.x), z0 = __builtin_amdgcn_readfirstlane(tv.y), y0 = __builtin_amdgcn_readfirstlane(tv.z), x0 = __builtin_amdgcn_readfirstlane(tv.w);
        {
            const long long sample = (long long)p.D * p.H * p.W * Cs;
            const __amdgpu_buffer_rsrc_t rs = da_rsrc_n<HB>(src, n, sample);
            const bool interior = smallA && z0 >= 1 && z0 + HZ - 2 < p.D && y0 >= 1 && y0 + HY - 2 < p.H && x0 >= 1 && x0 + HX - 2 < p.W;
            const unsigned Cs4 = (unsigned)Cs * ES, cofs4 = (unsigned)(choff + c4 * 4) * ES;      // (bytes per voxel record / of this thread's quad)
            const unsigned base = (unsigned)(((z0 - 1) * p.H + (y0 - 1)) * p.W + (x0 - 1)) * Cs4 + cofs4;
            if constexpr (PRO) vmA = 0;
#pragma unroll
            for (int it = 0; it < NITA; ++it) {
                const int hz = (int)(pkA[it] >> 16), hy = (int)((pkA[it] >> 8) & 255u), hx = (int)(pkA[it] & 255u);
                unsigned so;
                if (interior) so = ((it + 1) * NT <= TOTA || hz != 0xFFFF) ? __umul24((unsigned)voA[it], Cs4) + base : 0xFFFFFFFFu;
                else {
                    const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
                    const bool inb = (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                    so = inb ? (unsigned)(((z * p.H + y) * p.W + x)) * Cs4 + cofs4 : 0xFFFFFFFFu;
                }
                preA[it] = da_buf_loadq<HB, RAWA>(rs, so);
                if constexpr (PRO) vmA |= (so != 0xFFFFFFFFu ? 1u : 0u) << it;
            }
        }
        const long long sampleY = (long long)p.D * p.H * p.W * p.Cout;
        const __amdgpu_buffer_rsrc_t ry = da_rsrc_n<HB>(p.dy, n, sampleY);
        const bool inside = smallY && z0 + TZ <= p.D && y0 + TY <= p.H && x0 + TX <= p.W && cg * CG + CG <= p.Cout;
        const int basev = (z0 * p.H + y0) * p.W + x0;
        const unsigned baseY = ((unsigned)basev * (unsigned)p.Cout + (unsigned)yq4) * ES;
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
            const int v = yv0 + it * (NT / QY);
            const int vx = v & 15, vy = (v >> 4) & 7, vz = v >> 7;
            unsigned off;
            if (inside) off = __umul24((unsigned)voY[it], (unsigned)p.Cout * ES) + baseY;
            else {
                const int x = x0 + vx, y = y0 + vy, z = z0 + vz;
                const bool vin = z < p.D && y < p.H && x < p.W && yq4 < p.Cout;
                off = vin ? (unsigned)((((z * p.H + y) * p.W + x) * p.Cout + yq4) * ES) : 0xFFFFFFFFu;
            }
            preY[it] = da_buf_loadq<HB, RAWY>(ry, off);
        }
    };
    int Eacc = 0, Emin = 0, Enext = 0; bool first_tile = true;
    auto publish_max = [&]() {
        if constexpr (PRO) stage_pro_apply<0, NITA>(preA, vmA, psc, psf, pslope);
        if constexpr (SPL) {
            const float ma = da_wave_max_nonneg(stage_absmax<NITA>(preA)), my = da_wave_max_nonneg(stage_absmax<NITY>(preY));
            if (lane == 0) { smax[wave] = ma; smax[8 + wave] = my; }
        }
    };
    auto pack_bf16 = [&](const float4 v, bool raw) -> uint2 {             // one parked quad as four bf16 (raw: it already is, in .x / .y)
        return raw ? make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)) : make_uint2(da_bf16x2(v.x, v.y), da_bf16x2(v.z, v.w));
    };
    auto write_lds = [&]() {
        if constexpr (!SPL) {
#pragma unroll
            for (int it = 0; it < NITA; ++it) {
                const int idx0 = threadIdx.x + it * NT;
                if (idx0 < TOTA) {
                    const int hv = idx0 >> 2;
                    reinterpret_cast<uint2*>(ldsA)[(c4 >> 1) * (PLH / 4) + hv * 2 + (c4 & 1) + ZPQ * (hv / (HY * HX))] = pack_bf16(preA[it], RAWA);
                }
            }
#pragma unroll
            for (int it = 0; it < NITY; ++it) {
                const int idx = threadIdx.x + it * NT;
                if (idx < TVOX * QY) reinterpret_cast<uint2*>(ldsY)[idx] = pack_bf16(preY[it], RAWY);
            }
            return;
        }
        const float4 ma0 = *reinterpret_cast<const float4*>(smax), ma1 = *reinterpret_cast<const float4*>(smax + 4);
        const float4 my0 = *reinterpret_cast<const float4*>(smax + 8), my1 = *reinterpret_cast<const float4*>(smax + 12);
        const float mA = fmaxf(fmaxf(fmaxf(ma0.x, ma0.y), fmaxf(ma0.z, ma0.w)), fmaxf(fmaxf(ma1.x, ma1.y), fmaxf(ma1.z, ma1.w)));
        const float mY = fmaxf(fmaxf(fmaxf(my0.x, my0.y), fmaxf(my0.z, my0.w)), fmaxf(fmaxf(my1.x, my1.y), fmaxf(my1.z, my1.w)));
        const int ea = da_scale_exp(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mA))));
        const int ey = da_scale_exp(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mY))));
        int E = ea + ey;
        if (!first_tile) E = min(E, Emin + 40);
        if (!first_tile && E >= Eacc && E <= Eacc + 3) E = Eacc;           // (keep the accumulators' unit: see conv3_split_wgrad_kernel)
        Emin = first_tile ? E : min(Emin, E);
        first_tile = false;
        Enext = E;
        const float sy = da_pow2(ey), sa = da_pow2(E - ey);
#pragma unroll
        for (int it = 0; it < NITA; ++it) {
            const int idx0 = threadIdx.x + it * NT;
            if (idx0 < TOTA) {
                const int hv = idx0 >> 2;
                const int idx = (c4 >> 1) * (PLH / 4) + hv * 2 + (c4 & 1) + ZPQ * (hv / (HY * HX));
                uint2 h, l; da_split2(preA[it], sa, h, l);
                reinterpret_cast<uint2*>(ldsA)[idx] = h; reinterpret_cast<uint2*>(ldsA)[idx + PLA / 4] = l;
            }
        }
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
            const int idx = threadIdx.x + it * NT;
            if (idx < TVOX * QY) {
                uint2 h, l; da_split2(preY[it], sy, h, l);
                reinterpret_cast<uint2*>(ldsY)[idx] = h; reinterpret_cast<uint2*>(ldsY)[idx + PLY / 4] = l;
            }
        }
    };
    int4 tnext = fetch_tile(1);
    if (tw.cnt > 0) { issue_loads(fetch_tile(0)); publish_max(); __syncthreads(); write_lds(); }
    __syncthreads();
    constexpr int NPR = SPL ? 3 : 1;
    constexpr int PA[3] = {0, SPL ? 1 : 0, 0}, PB[3] = {SPL ? 1 : 0, 0, 0};      // (x, dY) plane pairs, small terms first (one plane: the single product)
#pragma unroll 1
    for (int tile = 0; tile < tw.cnt; ++tile) {
        const bool has_next = tile + 1 < tw.cnt;
        if (has_next && !(p.ablate & 1)) issue_loads(tnext);
        tnext = fetch_tile(tile + 2);
        if (SPL && Enext != Eacc) {
            const float f = da_acc_factor(Enext - Eacc);
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) acc[c][d] = acc[c][d] * f;
            Eacc = Enext;
        }
        if (!(p.ablate & 2)) {
        F3 Y0 = loadY(0), Y1 = loadY(1);
        F3 Fa = loadF(0, 0), Fb = loadF(0, 1), Fc = loadF(0, 2), Fd, Na, Nb, Nc;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            Fd = loadF(c, 3);
            Na = (c < 3) ? loadF(c + 1, 0) : loadG(0);
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                acc[c][0] = mma(acc[c][0], Fa.p[PA[pr]], Y0.p[PB[pr]]);
                acc[c][1] = mma(acc[c][1], Fb.p[PA[pr]], Y0.p[PB[pr]]);
                acc[c][2] = mma(acc[c][2], Fc.p[PA[pr]], Y0.p[PB[pr]]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (c < 3) { Nb = loadF(c + 1, 1); Nc = loadF(c + 1, 2); } else { Nb = loadG(1); Nc = loadF(4, 2); }
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                acc[c][0] = mma(acc[c][0], Fb.p[PA[pr]], Y1.p[PB[pr]]);
                acc[c][1] = mma(acc[c][1], Fc.p[PA[pr]], Y1.p[PB[pr]]);
                acc[c][2] = mma(acc[c][2], Fd.p[PA[pr]], Y1.p[PB[pr]]);
            }
            __builtin_amdgcn_sched_barrier(0);
            Fa = Na; Fb = Nb; Fc = Nc;
        }
        {
            Fd = loadF(4, 3);
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                acc[4][0] = mma(acc[4][0], Fa.p[PA[pr]], Y0.p[PB[pr]]);
                acc[4][1] = mma(acc[4][1], Fc.p[PA[pr]], Y0.p[PB[pr]]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                acc[4][0] = mma(acc[4][0], Fb.p[PA[pr]], Y1.p[PB[pr]]);
                acc[4][1] = mma(acc[4][1], Fd.p[PA[pr]], Y1.p[PB[pr]]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (has_next && !(p.ablate & 4)) {
            publish_max();
            __syncthreads();
            write_lds();
            __syncthreads();
        }
    }
    if constexpr (SPL) {
        const float inv1 = da_pow2(-(Eacc / 2)), inv2 = da_pow2(-(Eacc - Eacc / 2));
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) acc[c][d] = acc[c][d] * inv1 * inv2;
    }
    // reduce the four row-pair waves of each channel half through LDS (two rounds; 4 x 15 KB), then waves 0 and 4 write the slab's partial dW
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(lds);
    auto put = [&](int slot) {
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) red[((slot * 15) + c * 3 + d) * 64 + lane] = make_float4(acc[c][d][0], acc[c][d][1], acc[c][d][2], acc[c][d][3]);
    };
    auto add = [&](int slot) {
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float4 v = red[((slot * 15) + c * 3 + d) * 64 + lane];
                acc[c][d][0] += v.x; acc[c][d][1] += v.y; acc[c][d][2] += v.z; acc[c][d][3] += v.w;
            }
    };
    if (wr >= 2) put(2 * wh + wr - 2);
    __syncthreads();
    if (wr < 2) add(2 * wh + wr);
    __syncthreads();
    if (wr == 1) put(wh);
    __syncthreads();
    if (wr == 0) {
        add(wh);
        float* part = p.partial + (size_t)slab * p.O;
        const int Cin = p.C1 + p.C2;
        const int co = cg * CG + i;
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 4 * g + reg;
                    const int combo = c < 4 ? 2 * c + (row >> 3) : 8;
                    const int dyt = c < 4 ? d : 2 * d + (row >> 3);
                    const int tap = (combo / 3) * 9 + dyt * 3 + combo % 3, ci = wh * 8 + (row & 7);
                    if (dyt < 3 && (c < 4 || d < 2) && co < p.Cout) part[((size_t)tap * Cin + cbase + ci) * p.Cout + co] = acc[c][d][reg];
                }
    }
}

#include "conv3d_wgring.h"      // conv3_split_wgrad16r_kernel: the eight-wave form walking z columns with the x planes in an LDS ring

__global__ void slab_reduce_kernel(const float* __restrict__ partial, int nparts, int O, float* __restrict__ out) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < O; o += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int b = 0; b < nparts; ++b) s += (double)partial[(size_t)b * O + o];
        out[o] = (float)s;
    }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient for tiny Cin (first layers: seg 1 -> 8, reg 2 -> 16).  HBM-bound reduction over voxels:
// GEMM M = (tap, ci) flattened (27*Cin <= 112 -> MT tiles of 16), N = Cout, K = voxels; both operands are read
// straight from global memory in fragment order (the 27 shifted input reads hit L1/L2), one wave per output row.
// ---------------------------------------------------------------------------------------------------
struct ScP {
    const float* in1; const float* in2; int C1, C2;
    const float* dy; float* partial;
    int N, D, H, W, Cout; long long nrows;
    const float* dyb; int Cd1;      // optional second half of the B operand: channels [Cd1, Cout) come from dyb (stride Cout - Cd1);
                                    // Cd1 == Cout when the B operand is one tensor
};

template <int MT, int NT>
__global__ void __launch_bounds__(256) conv3_smallcin_wgrad_kernel(ScP p) {
    __shared__ __attribute__((aligned(16))) float red[MT * NT * 64 * 4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int Cin = p.C1 + p.C2;
    // per-lane description of the A rows this lane feeds: m = 16*mt + i -> (tap, ci)
    int dz[MT], dy_[MT], dx[MT], cs[MT], cc[MT]; bool mval[MT]; const float* src[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 16 * mt + i;
        mval[mt] = m < 27 * Cin;
        const int tap = mval[mt] ? m / Cin : 0, ci = mval[mt] ? m % Cin : 0;
        dz[mt] = tap / 9 - 1; dy_[mt] = (tap / 3) % 3 - 1; dx[mt] = tap % 3 - 1;
        if (ci < p.C1) { src[mt] = p.in1; cs[mt] = p.C1; cc[mt] = ci; } else { src[mt] = p.in2; cs[mt] = p.C2; cc[mt] = ci - p.C1; }
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Operands come through buffer descriptors with 32-bit byte offsets (out-of-range lane -> 0xFFFFFFFF -> 0): no
    // exec-mask branches, so the MT + NT loads of a K-step issue back to back, and the next step's loads are in flight
    // while this step's MFMAs issue.
    // Descriptors are per SAMPLE (rebuilt when the wave's row moves to another sample), so only one sample has to fit 32 bits.
    const unsigned long long vol = (unsigned long long)p.D * p.H * p.W;
    const int Cd2 = p.Cout - p.Cd1;
    bool from2[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) from2[mt] = (src[mt] == p.in2) && p.C2 > 0;
    for (DaXcdItems RL = da_xcd_items(p.nrows, wave, 4); RL.i < RL.end; RL.i += RL.step) {      // (rows of one XCD are neighbours: the 9 (dz, dy) uses of an input row meet in its L2)
        const long long row = RL.i;
        const int y = (int)(row % p.H); const int z = (int)((row / p.H) % p.D);
        const int n = __builtin_amdgcn_readfirstlane((int)(row / ((long long)p.H * p.D)));
        const __amdgpu_buffer_rsrc_t r1 = da_rsrc(p.in1 + (size_t)n * vol * p.C1, (unsigned)(vol * p.C1 * 4ull));
        const __amdgpu_buffer_rsrc_t r2 = da_rsrc(p.C2 > 0 ? p.in2 + (size_t)n * vol * p.C2 : p.in1, (unsigned)(vol * (p.C2 > 0 ? p.C2 : p.C1) * 4ull));
        const __amdgpu_buffer_rsrc_t ry = da_rsrc(p.dy + (size_t)n * vol * p.Cd1, (unsigned)(vol * p.Cd1 * 4ull));
        const __amdgpu_buffer_rsrc_t ry2 = da_rsrc(Cd2 > 0 ? p.dyb + (size_t)n * vol * Cd2 : p.dy, (unsigned)(vol * (Cd2 > 0 ? Cd2 : p.Cd1) * 4ull));
        unsigned rbase[MT]; bool rval[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int zz = z + dz[mt], yy = y + dy_[mt];
            rval[mt] = mval[mt] && zz >= 0 && zz < p.D && yy >= 0 && yy < p.H;
            rbase[mt] = (unsigned)((((((long long)(rval[mt] ? zz : 0)) * p.H + (rval[mt] ? yy : 0)) * p.W) * cs[mt] + cc[mt]) * 4);
        }
        const long long srow = (long long)z * p.H + y;                 // row within the sample
        const unsigned gbase = (unsigned)((srow * p.W) * p.Cd1 * 4), gbase2 = (unsigned)((srow * p.W) * Cd2 * 4);
        auto fetch = [&](int x0, float* a, float* b) {
            const int x = x0 + g;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = 16 * nt + i;
                const bool ok = x < p.W && c < p.Cout;
                if (c < p.Cd1) b[nt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, ok ? gbase + (unsigned)((x * p.Cd1 + c) * 4) : 0xFFFFFFFFu, 0, 0));
                else b[nt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry2, ok ? gbase2 + (unsigned)((x * Cd2 + c - p.Cd1) * 4) : 0xFFFFFFFFu, 0, 0));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int xx = x + dx[mt];
                const unsigned off = (rval[mt] && x < p.W && xx >= 0 && xx < p.W) ? rbase[mt] + (unsigned)(xx * cs[mt] * 4) : 0xFFFFFFFFu;
                a[mt] = __builtin_bit_cast(float, from2[mt] ? __builtin_amdgcn_raw_buffer_load_b32(r2, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b32(r1, off, 0, 0));
            }
        };
        float aA[MT], bA[NT], aB[MT], bB[NT];
        fetch(0, aA, bA);
#pragma unroll 1
        for (int x0 = 0; x0 < p.W; x0 += 8) {
            fetch(x0 + 4, aB, bB);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aA[mt], bA[nt], acc[mt][nt], 0, 0, 0);
            fetch(x0 + 8, aA, bA);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aB[mt], bB[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float4* slot = reinterpret_cast<float4*>(red) + (mt * NT + nt) * 64 + lane;
                    float4 cur = (w == 0) ? make_float4(0.f, 0.f, 0.f, 0.f) : *slot;
                    cur.x += acc[mt][nt][0]; cur.y += acc[mt][nt][1]; cur.z += acc[mt][nt][2]; cur.w += acc[mt][nt][3];
                    *slot = cur;
                }
        }
        __syncthreads();
    }
    const int O = 27 * Cin * p.Cout;
    float* part = p.partial + (size_t)blockIdx.x * O;
    for (int idx = threadIdx.x; idx < MT * NT * 64; idx += 256) {
        const int ln = idx & 63, q = idx >> 6;
        const int nt = q % NT, mt = q / NT;
        const float4 v = reinterpret_cast<const float4*>(red)[idx];
        const float vals[4] = {v.x, v.y, v.z, v.w};
        const int co = 16 * nt + (ln & 15);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = 16 * mt + 4 * (ln >> 4) + reg;
            if (m < 27 * Cin && co < p.Cout) part[(size_t)m * p.Cout + co] = vals[reg];
        }
    }
}

// Tap masks of the stride-1 conv over the space-to-depth tensor (channel = parity*Cin + ci, parity = (rz*2+ry)*2+rx).
// Along an axis an even-parity (r=0) sub-volume only meets the original centre tap (offset index 1); an odd one meets
// taps 0 and 2 at offset indices 0 and 1.  groups[g] covers channels [g*gsize, (g+1)*gsize); flipped: bit 26 - t.
static void da_s2d_masks(unsigned* masks, int ngroups, int gsize, int cin, int flipped) {
    for (int g = 0; g < 16; ++g) masks[g] = 0;
    for (int g = 0; g < ngroups && g < 16; ++g) {
        const int r0 = (g * gsize) / cin, r1 = ((g + 1) * gsize - 1) / cin;
        unsigned m = 0;
        for (int r = r0; r <= r1 && r < 8; ++r) {
            const int rz = (r >> 2) & 1, ry = (r >> 1) & 1, rx = r & 1;
            for (int oz = (rz ? 0 : 1); oz <= 1; ++oz)
                for (int oy = (ry ? 0 : 1); oy <= 1; ++oy)
                    for (int ox = (rx ? 0 : 1); ox <= 1; ++ox) {
                        const int t = (oz * 3 + oy) * 3 + ox;
                        m |= 1u << (flipped ? 26 - t : t);
                    }
        }
        masks[g] = m;
    }
}

static int pick_ck(int C1, int C2) {
    const int Cin = C1 + C2;
    if (Cin % 16 == 0 && C1 % 16 == 0) return 16;
    if (Cin % 8 == 0 && C1 % 8 == 0) return 8;
    return 0;
}
// N-tiles per workgroup: <= 2, so that 8*NREP*4 accumulators + fragments + the register-parked staging prefetch stay
// under 256 VGPRs (two workgroups per CU); more couts go to blockIdx.y (the input tile is re-staged per group, which the
// one-item-ahead prefetch hides).
static int pick_nrep(int NT) { return NT <= 3 ? NT : (NT % 2 == 0 ? 2 : (NT % 3 == 0 ? 3 : 2)); }

static const int kDynCtrInts = 256;          // tile counters [<= 32 cout groups][8 XCDs] behind the packed weights; split mode: the chunks' weight exponents (<= 256 chunks)
static size_t packed_bytes(int Cin, int Cout, int CK) {
    const int NT = (Cout + 15) / 16, NREP = pick_nrep(NT);
    const int NTpad = ((NT + NREP - 1) / NREP * NREP + 1) & ~1;      // (even: the bf16 / split modes use <= 2 N-tiles per workgroup)
    const int NSTEPS = (27 * CK + 15) / 16;
    size_t b = (size_t)(Cin / CK) * NSTEPS * NTpad * 256 * sizeof(float);
    const size_t sp = (size_t)(Cin / 8) * 7 * NTpad * 2048;           // split mode: 8-channel chunks, 7 K-steps of 32, two 1 KiB planes
    if (Cin % 8 == 0 && sp > b) b = sp;
    return da_align(b) + da_align(kDynCtrInts * sizeof(int));
}

struct WgPlan { int CK, NREP, ngroups, nchunks, ntz, nty, ntx, ntiles, nslabs, tps; size_t partial_bytes; int w16; };      // w16: 1 conv3_split_wgrad16_kernel (16-channel chunks, one 8-wave workgroup per CU), 2 its ring form conv3_split_wgrad16r_kernel
static WgPlan wgrad_plan(int N, int D, int H, int W, int C1, int C2, int Cout, bool split = false, bool allow16 = false, bool allow_ring = false) {
    WgPlan q;
    q.CK = pick_ck(C1, C2);
    const int NT = (Cout + 15) / 16;
    q.NREP = NT >= 2 ? 2 : 1;
    if (split && q.CK) { q.CK = 8; q.NREP = 1; }             // split mode: two fp16 planes of x and dY in LDS -> 8-channel chunks, one cout tile
    q.ngroups = (NT + q.NREP - 1) / q.NREP;
    q.nchunks = q.CK ? (C1 + C2) / q.CK : 1;
    q.ntz = (split && q.CK) ? (D + DA_WG_TZ - 1) / DA_WG_TZ : (D + 1) / 2; q.nty = (H + TY - 1) / TY; q.ntx = (W + TX - 1) / TX;      // (the row-owner kernel: DA_WG_TZ planes per tile)
    q.ntiles = N * q.ntz * q.nty * q.ntx;
    const size_t O = (size_t)27 * (C1 + C2) * Cout;
    long long slabs = 512 / (q.nchunks * q.ngroups); if (slabs < 1) slabs = 1;      // one resident round: 2 workgroups / CU
    const long long cap = (long long)((96ull << 20) / (O * 4)); if (slabs > cap) slabs = cap < 1 ? 1 : cap;
    q.w16 = 0;
    if (allow16 && split && q.CK && C1 % 16 == 0 && C2 % 16 == 0 && DA_WG_TZ == 2) {
        static int on = -1; if (on < 0) { const char* e = getenv("DA_WG16"); on = (e && !atoi(e)) ? 0 : 1; }
        const int combos = ((C1 + C2) / 16) * q.ngroups;
        long long s16 = (256 / combos) & ~7ll;               // one workgroup per CU, a multiple of 8 slabs (XCD grouping) ...
        { static int any = -1; if (any < 0) { const char* e = getenv("DA_WG16_ANY"); any = (e && !atoi(e)) ? 0 : 1; }
          if (any && s16 * combos < 224) s16 = 256 / combos; }  // ... or any slab count that fills the chip (tile_walk then walks one list): 192 -> 64 0.65 -> 0.60 ms, 96 -> 32 unchanged
        if (on && s16 >= 1 && s16 * combos >= 224 && s16 <= cap && s16 <= q.ntiles) { q.w16 = 1; q.nchunks = (C1 + C2) / 16; slabs = s16; }
        // the ring form (conv3d_wgring.h): fp32 tensors in split mode only; contiguous tile ranges per slab, so no multiple-of-8 rounding below
        static int ring = -1; if (ring < 0) { const char* e = getenv("DA_WG16R"); ring = (e && !atoi(e)) ? 0 : 1; }
        if (q.w16 && ring && allow_ring) {
            q.w16 = 2;
            // DA_WG16R_SLABMUL=k: k times as many (k times shorter) workgroups than CUs can hold at once, so that workgroups of a kernel queued later
            // on a higher-priority stream get CUs as these retire (experiment: the persistent form holds every CU until its slab is done)
            static int mul = -1; if (mul < 0) { const char* e = getenv("DA_WG16R_SLABMUL"); mul = e ? atoi(e) : 1; if (mul < 1) mul = 1; }
            slabs *= mul; if (slabs > cap) slabs = cap;
            if (slabs > q.ntiles) slabs = q.ntiles;
            q.tps = (int)da_cdiv(q.ntiles, slabs);
            q.nslabs = (int)da_cdiv(q.ntiles, q.tps);           // (every slab non-empty)
            q.partial_bytes = da_align((size_t)q.nslabs * O * sizeof(float));
            return q;
        }
    }
    if (slabs > q.ntiles) slabs = q.ntiles;
    if (slabs >= 8) slabs &= ~7ll;                           // multiple of 8: blockIdx.x % 8 is then the XCD (tile_walk)
    q.tps = (int)da_cdiv(q.ntiles, slabs);
    q.nslabs = (int)slabs;
    q.partial_bytes = da_align((size_t)q.nslabs * O * sizeof(float));
    return q;
}

}  // namespace

static size_t tile_table_bytes(int N, int D, int H, int W) {
    return da_align((size_t)N * ((D + 3) / 4) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX) * sizeof(int4));
}
static size_t wg_tile_table_bytes(int N, int D, int H, int W) {
    return da_align((size_t)N * ((D + 1) / 2) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX) * sizeof(int4));
}
static const int kScBlocksFwd = 1024;
// layout of a forward / data-gradient call: [packed weights | tile counters][tile table]; of a weight-gradient call: [partials]
size_t da_conv3_mfma_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int stride) {
    if (stride != 1) return 0;
    size_t pk = 0;
    if (Cin % 8 == 0) { const size_t a = packed_bytes(Cin, Cout, Cin % 16 == 0 ? 16 : 8); if (a > pk) pk = a; const size_t b = packed_bytes(Cin, Cout, 8); if (b > pk) pk = b; }
    if (Cout % 8 == 0) { const size_t a = packed_bytes(Cout, Cin, Cout % 16 == 0 ? 16 : 8); if (a > pk) pk = a; }
    size_t part = 0;
    if (Cin % 8 == 0) { part = wgrad_plan(N, D, H, W, Cin, 0, Cout).partial_bytes; const size_t pr = wgrad_plan(N, D, H, W, Cin, 0, Cout, true, true, true).partial_bytes; if (pr > part) part = pr; }
    if (Cin <= 4 && Cout <= 32) part = da_align((size_t)kScBlocksFwd * 27 * Cin * Cout * sizeof(float));
    if (Cout <= 4 && Cin <= 32) { const size_t sw = da_align((size_t)(kScBlocksFwd + 1) * 27 * Cin * Cout * sizeof(float)); if (sw > part) part = sw; }   // swapped-operand weight gradient
    return pk + tile_table_bytes(N, D, H, W) + part + wg_tile_table_bytes(N, D, H, W);
}

bool da_conv3_mfma_fwd_supported(int C1, int C2, int Cout, int stride, int Cs1, int Cs2) {
    if (stride != 1) return false;
    if (pick_ck(C1, C2) == 0) return false;
    if (Cout < 8) return false;            // Cout = 3 (flow) wastes 13/16 of every MFMA: direct kernel is faster
    // 16-byte epilogue stores through one buffer descriptor per N-tile: couts in quads, a split output on a tile boundary
    if (Cout % 4 != 0) return false;
    if (Cs1 < 0) { Cs1 = Cout; Cs2 = 0; }
    if (Cs2 > 0 && (Cs1 % 16 != 0 || Cs2 % 4 != 0)) return false;
    if ((unsigned long long)Cs1 % 4 != 0) return false;
    return true;
}

template <int CK, int NREP, bool MASKED = false, int STATS = 0, bool BF = false, bool PRO = false, bool DYN = false, bool SP = false, int S2F = 0, bool PAIR = false, bool HB = false, int WPE = 2>
static int launch_fwd_mfma(const FwdP& p, int gy, hipStream_t st) {
    const size_t shm = (size_t)6 * HY * HX * CK * (BF ? 2 : 4) * (SP ? 2 : 1) + (STATS ? (size_t)4 * 2 * NREP * 16 * sizeof(double) : 0) + ((DYN || SP) ? 16 : 0);
    auto kern = conv3_mfma_fwd_kernel<CK, NREP, MASKED, STATS, BF, PRO, DYN, SP, S2F, PAIR, HB, WPE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    { static int occ = -1; if (occ < 0) { occ = getenv("DA_OCC") ? 1 : 0; if (occ) { int nb = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, shm); fprintf(stderr, "[occ] <%d,%d,M%d,S%d,BF%d,PRO%d,DYN%d,SP%d> lds %zu B -> %d workgroups per CU\n", CK, NREP, (int)MASKED, (int)STATS, (int)BF, (int)PRO, (int)DYN, (int)SP, shm, nb); } } }
    static unsigned long long* dclk = nullptr; static int want = -1;
    if (want < 0) { want = getenv("DA_CLK") ? 1 : 0; if (want) (void)hipMalloc(&dclk, 64 + 1024 * 16); }
    FwdP q = p; q.clk = want ? dclk : nullptr;
    if (want) { const unsigned long long init[6] = {0, 0, ~0ull, 0, ~0ull, 0}; (void)hipMemcpyAsync(dclk, init, 48, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }
    hipLaunchKernelGGL(kern, dim3(p.nblocks, gy), dim3(256), shm, st, q);
    if (want) {
        unsigned long long h[6]; (void)hipStreamSynchronize(st); (void)hipMemcpy(h, dclk, 48, hipMemcpyDeviceToHost);
        fprintf(stderr, "[clk] cycles %llu realtime %llu -> %.0f MHz; grid %d x %d: starts span %.1f us, ends span %.1f us, first start -> last end %.1f us\n", h[0], h[1],
                (double)h[0] / (double)h[1] * 100.0, p.nblocks, gy, (h[3] - h[2]) * 0.01, (h[5] - h[4]) * 0.01, (h[5] - h[2]) * 0.01);
        if (getenv("DA_CLK_DUMP")) {
            static unsigned long long rec[2048];
            const int nb = p.nblocks < 1024 ? p.nblocks : 1024;
            (void)hipMemcpy(rec, dclk + 8, (size_t)nb * 16, hipMemcpyDeviceToHost);
            for (int b = 0; b < nb; ++b) fprintf(stderr, "[blk] %d xcc %llu hwid %08llx life_us %.1f\n", b, (rec[2 * b] >> 32) & 15, rec[2 * b] & 0xffffffffull, rec[2 * b + 1] * 0.01);
        }
    }
    DA_LAUNCH_CHECK();
    return 0;
}

// identity scale / shift for an input without a prologue (the kernels load the constants unconditionally); one device per process
static const int kProMaxC = 2048;
static int pro_identity(const float** ones, const float** zeros) {
    static float* buf = nullptr;
    if (!buf) {
        hipError_t e = hipMalloc(&buf, 2 * kProMaxC * sizeof(float));
        if (e != hipSuccess) return (int)e;
        float* h = (float*)malloc(2 * kProMaxC * sizeof(float));
        for (int i = 0; i < kProMaxC; ++i) { h[i] = 1.f; h[kProMaxC + i] = 0.f; }
        e = hipMemcpy(buf, h, 2 * kProMaxC * sizeof(float), hipMemcpyHostToDevice);
        free(h);
        if (e != hipSuccess) return (int)e;
    }
    *ones = buf; *zeros = buf + kProMaxC;
    return 0;
}

// kernel-side activation parameter (da_act01): slope in [0, 1) as is, "no activation" (slope < 0 or no prologue) -> 1
static int pro_slopes(const DaPro* pro, int C2, float* s1, float* s2) {
    const float a = pro->s1 ? pro->slope1 : -1.f, b = (C2 > 0 && pro->s2) ? pro->slope2 : -1.f;
    if (a >= 1.f || b >= 1.f) return 1;
    *s1 = a < 0.f ? 1.f : a; *s2 = b < 0.f ? 1.f : b;
    return 0;
}

static int conv3_mfma_fwd_impl(const float* in1, int C1, const float* in2, int C2, const float* w_tio, int w_is_flipped_tr,
                               const float* bias, float* out1, int Cs1, float* out2, int Cs2,
                               int N, int D, int H, int W, int Cout, int stride, float slope,
                               void* ws, size_t ws_bytes, hipStream_t st, int s2d_cin, double* stats_partial, int* stats_nparts, const DaPro* pro, const DaS2dFuse* s2f,
                               int cout0, int CoutW, bool hb);

// ---- packed operands kept across calls (split mode) -----------------------------------------------------------------------------------------
// A convolution call packs its weights (fragment order, two fp16 planes, per-chunk exponents) and writes its tile table before the matrix kernel can
// start: a 10-us launch in the dependent chain of every layer, forward and data gradient, every step, for data that only changes when the optimiser
// steps.  The caller may keep the packed operand itself: da_conv3d_k3_prepack fills a caller-owned buffer (the host side does so for every layer right
// after the optimiser step, on the side stream), da_conv3d_k3_use_prepacked hands it to the NEXT convolution call of this thread on the same weights.
// mode 1: use, 2: fill (the matrix kernel is not launched).  Up to two regions: the 48 <- 16 data gradient runs two launches with their own packs.
// BatchNorm-backward sums in the data gradient's epilogue (STATS == 2): set by da_conv3d_k3_dgrad_bst for the one call it makes
struct BstState { const float* y; const float* par; float slope; };
static thread_local BstState g_bst = {nullptr, nullptr, -1.f};

struct PrepackState { int mode; const float* w; char* buf[2]; size_t bytes[2]; int next, used; };
static thread_local PrepackState g_pp = {0, nullptr, {nullptr, nullptr}, {0, 0}, 0, 0};
static thread_local PackJobs* g_pp_collect = nullptr;       // fill mode: record the pack launches here instead of issuing them (da_conv3d_k3_prepack_many)
static thread_local int g_pp_ncollect = 0;

int da_conv3_mfma_fwd(const float* in1, int C1, const float* in2, int C2, const float* w_tio, int w_is_flipped_tr,
                      const float* bias, float* out1, int Cs1, float* out2, int Cs2,
                      int N, int D, int H, int W, int Cout, int stride, float slope,
                      void* ws, size_t ws_bytes, hipStream_t st, int s2d_cin, double* stats_partial, int* stats_nparts, const DaPro* pro, const DaS2dFuse* s2f, int act_bf16) {
    if (act_bf16 && da_matrix_mode() != 1) return DA_ERR_UNSUPPORTED;          // bf16 activation storage goes with the bf16 matrix mode
    // Split mode, data gradient of a concat layer whose outputs are 32 + 16 channels (the 48 -> 16 decoder convolution: three N-tiles).  One
    // launch with one N-tile per workgroup stages dY three times; two launches -- two N-tiles sharing every dY fragment for the first output
    // tensor, one N-tile (paired staging) for the second -- stage it twice and write each output tensor from its own launch.
    struct PpReset { ~PpReset() { if (g_pp.mode == 1) g_pp.mode = 0; } } pp_reset;      // a handed-over pack serves exactly one call
    if (g_pp.mode && g_pp.w != w_tio) g_pp.mode = 0;                                      // (it belongs to other weights: a call in between went elsewhere)
    static int no2 = -1; if (no2 < 0) { const char* e = getenv("DA_NO_DGRAD_SPLIT_LAUNCH"); no2 = (e && atoi(e)) ? 1 : 0; }
    if (!no2 && da_matrix_mode() == 2 && w_is_flipped_tr && s2d_cin == 0 && !stats_partial && !pro && Cs2 > 0 && Cs1 == 32 && Cs2 == 16 && Cout == 48 &&
        pick_ck(C1, C2) != 0) {
        int rc = conv3_mfma_fwd_impl(in1, C1, in2, C2, w_tio, 1, bias, out1, 32, nullptr, 0, N, D, H, W, 32, stride, slope, ws, ws_bytes, st, 0, nullptr, nullptr,
                                     nullptr, nullptr, 0, Cout, false);
        if (rc) return rc;
        return conv3_mfma_fwd_impl(in1, C1, in2, C2, w_tio, 1, bias ? bias + 32 : nullptr, out2, 16, nullptr, 0, N, D, H, W, 16, stride, slope, ws, ws_bytes, st, 0,
                                   nullptr, nullptr, nullptr, nullptr, 32, Cout, false);
    }
    return conv3_mfma_fwd_impl(in1, C1, in2, C2, w_tio, w_is_flipped_tr, bias, out1, Cs1, out2, Cs2, N, D, H, W, Cout, stride, slope, ws, ws_bytes, st, s2d_cin,
                               stats_partial, stats_nparts, pro, s2f, 0, Cout, act_bf16 != 0);
}

static int conv3_mfma_fwd_impl(const float* in1, int C1, const float* in2, int C2, const float* w_tio, int w_is_flipped_tr,
                               const float* bias, float* out1, int Cs1, float* out2, int Cs2,
                               int N, int D, int H, int W, int Cout, int stride, float slope,
                               void* ws, size_t ws_bytes, hipStream_t st, int s2d_cin, double* stats_partial, int* stats_nparts, const DaPro* pro, const DaS2dFuse* s2f,
                               int cout0, int CoutW, bool hb) {
    (void)stride;
    const int Cin = C1 + C2;
    int CK = pick_ck(C1, C2);
    if (!CK) return DA_ERR_UNSUPPORTED;
    if (pro && (s2d_cin > 0 || w_is_flipped_tr || Cin > kProMaxC)) return DA_ERR_UNSUPPORTED;
    const bool split = da_matrix_mode() == 2 && s2d_cin == 0;      // (the sparse-tap stride-2 route keeps the native fp32 kernels)
    const bool bf = da_matrix_mode() == 1;
    if (split) CK = 8;
    {   // every tensor is addressed per sample through a buffer descriptor with 32-bit byte offsets
        const unsigned long long vox4 = (unsigned long long)D * H * W * 4ull;
        const int cmax = (C1 > C2 ? C1 : C2) > (Cs1 > Cs2 ? Cs1 : Cs2) ? (C1 > C2 ? C1 : C2) : (Cs1 > Cs2 ? Cs1 : Cs2);
        if (vox4 * (unsigned long long)cmax >= 0xFFFFFFF0ull) return DA_ERR_UNSUPPORTED;
    }
    const int NT = (Cout + 15) / 16;
    int NREP = pick_nrep(NT);
    if ((bf || split || pro) && NREP > 2) NREP = 2;        // bf16 mode is not matrix-bound, three N-tiles would spill; the prologue variant parks the whole next tile in registers
    const bool want_bst = g_bst.y != nullptr && split && stats_partial != nullptr;
    if (want_bst) NREP = 1;                                // the BatchNorm-backward epilogue holds TY quads of the producer's output: one N-tile per workgroup
    {   // Coarse levels have few tiles (30 per volume at 20x24x20, 180 at 40x48x40): the persistent grid then runs one or two
        // uneven rounds.  Makespan model: a workgroup walks ceil(tiles / nblk) tiles, each costing ~NREP (one N-tile per
        // workgroup is ~8 % less efficient per FLOP but quadruples / doubles the number of work items); take the cheaper.
        static int adapt = -1; if (adapt < 0) { const char* e = getenv("DA_NREP_ADAPT"); adapt = e ? atoi(e) : 1; }
        const long long tiles = (long long)N * ((D + 3) / 4) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX);
        if (adapt && s2d_cin == 0 && NREP > 1 && !want_bst) {
            auto cost = [&](int nrep) {
                const int g = (NT + nrep - 1) / nrep;
                long long nb = 512 / g; if (nb < 1) nb = 1; if (nb > tiles) nb = tiles;
                return (double)((tiles + nb - 1) / nb) * nrep * (nrep == 1 ? 1.08 : 1.0);
            };
            if (cost(1) < cost(NREP)) NREP = 1;
        }
    }
    const int gy = (NT + NREP - 1) / NREP, NTpad = gy * NREP;
    const int pkmode = split ? 3 : bf ? (s2d_cin > 0 ? 1 : 2) : 0;       // 0 fp32 | 1 bf16, one tap per K-step (sparse taps) | 2 bf16, K = 32 per step | 3 split: three bf16 planes, K = 32
    const int NSTEPS = pkmode >= 2 ? (CK == 16 ? 14 : 7) : (27 * CK + 15) / 16;
    const size_t pk = packed_bytes(Cin, Cout, CK);
    char* opbase = reinterpret_cast<char*>(ws);                 // [packed weights | exponents / tile counters][tile table]: the workspace, or the caller's kept copy
    int pp_mode = 0;
    if (g_pp.mode && split) {
        const int idx = g_pp.next++;
        if (idx < 2 && g_pp.buf[idx] && g_pp.bytes[idx] >= pk + tile_table_bytes(N, D, H, W)) { opbase = g_pp.buf[idx]; pp_mode = g_pp.mode; }
        else if (g_pp.mode == 2) return DA_ERR_WS_SMALL;
    } else if (g_pp.mode == 2) return DA_ERR_UNSUPPORTED;       // (nothing to keep outside the split mode)
    if (!pp_mode && ws_bytes < pk + tile_table_bytes(N, D, H, W)) return DA_ERR_WS_SMALL;
    float* wp = reinterpret_cast<float*>(opbase);
    int4* tiles = reinterpret_cast<int4*>(opbase + pk);
    const long long total = (long long)(Cin / CK) * NSTEPS * NTpad * 256;
    int* dyn_ctr = reinterpret_cast<int*>(opbase + pk - da_align(kDynCtrInts * sizeof(int)));
    static int dyn_env = -1; if (dyn_env < 0) { const char* e = getenv("DA_DYN_TILES"); dyn_env = (e && atoi(e)) ? 1 : 0; }
    const bool dyn = dyn_env && !bf && !split && !pro && s2d_cin == 0 && !stats_partial && gy * 8 <= kDynCtrInts;
    FwdP p;
    p.bst_y = nullptr; p.bst_par = nullptr; p.bst_slope = -1.f;
    p.ntz = (D + 3) / 4; p.nty = (H + TY - 1) / TY; p.ntx = (W + TX - 1) / TX;
    p.ntiles = N * p.ntz * p.nty * p.ntx;
    if (split) {
        if (Cin / 8 > kDynCtrInts) return DA_ERR_UNSUPPORTED;
        if (pp_mode == 2 && g_pp_collect) {
            if (g_pp_ncollect >= kPackJobsMax) return DA_ERR_WS_SMALL;
            g_pp_collect->j[g_pp_ncollect++] = PackJob{w_tio, reinterpret_cast<unsigned short*>(wp), dyn_ctr, tiles, Cin, Cout, NTpad, w_is_flipped_tr, p.ntiles, p.ntx, p.nty, p.ntz, cout0, CoutW};
        } else if (pp_mode != 1)
        hipLaunchKernelGGL(pack_split_weights_kernel, dim3(Cin / 8, 8), dim3(256), 0, st, w_tio, reinterpret_cast<unsigned short*>(wp), dyn_ctr, Cin, Cout, NTpad, w_is_flipped_tr,
                           tiles, p.ntiles, p.ntx, p.nty, p.ntz, cout0, CoutW);
        if (pp_mode == 2) { DA_LAUNCH_CHECK(); g_pp.used++; return 0; }
    } else
    hipLaunchKernelGGL(pack_fwd_weights_kernel, dim3(da_grid(total > p.ntiles ? total : p.ntiles, 256, 1024)), dim3(256), 0, st, w_tio, wp, Cin, Cout, CK, NSTEPS, NTpad, w_is_flipped_tr, total, pkmode,
                       dyn ? dyn_ctr : nullptr, gy * 8, tiles, p.ntiles, p.ntx, p.nty, p.ntz, cout0, CoutW);
    DA_LAUNCH_CHECK();
    p.tiles = tiles;
    p.in1 = in1; p.in2 = in2; p.C1 = C1; p.C2 = C2; p.wp = wp; p.bias = bias;
    p.out1 = out1; p.out2 = out2; p.Cs1 = Cs1; p.Cs2 = Cs2;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cout = Cout; p.NT = NTpad;
    p.ntz = (D + 3) / 4; p.nty = (H + TY - 1) / TY; p.ntx = (W + TX - 1) / TX;
    p.ntiles = N * p.ntz * p.nty * p.ntx; p.slope = slope;
    p.maskmode = 0;
    if (s2d_cin > 0) {
        if (CK != 16) return DA_ERR_UNSUPPORTED;
        if (!w_is_flipped_tr) { p.maskmode = 1; da_s2d_masks(p.masks, Cin / 16, 16, s2d_cin, 0); }
        else { p.maskmode = 2; da_s2d_masks(p.masks, gy, NREP * 16, s2d_cin, 1); }
    }
    { static int abl = -1; if (abl < 0) { const char* e = getenv("DA_ABLATE"); abl = e ? atoi(e) : 0; } p.ablate = abl; }
    {   // one resident round: 2 workgroups per CU x 256 CUs, split over the cout groups
        // (a third workgroup per CU for the split kernels -- 168 VGPRs, 3 x 52 KB of LDS -- was measured and dropped: in split mode the
        // matrix pipe is already busy ~100 % of the shader cycles and the clock is set by the power limit)
        static int nres = -1; if (nres < 0) { const char* e = getenv("DA_FWD_BLOCKS"); nres = e ? atoi(e) : 512; }
        static int wg3 = -1; if (wg3 < 0) { const char* e = getenv("DA_FWD_WG3"); wg3 = (e && atoi(e)) ? 1 : 0; }
        int nblk = ((wg3 && split && NREP == 1) ? 768 : nres) / gy; if (nblk < 1) nblk = 1; if (nblk > p.ntiles) nblk = p.ntiles;
        if (nblk >= 8) nblk &= ~7;                           // multiple of 8: blockIdx.x % 8 is then the XCD (tile_walk)
        p.nblocks = nblk;
    }
    { static int norot = -1; if (norot < 0) { const char* e = getenv("DA_PRIO_ROT"); norot = (e && atoi(e)) ? 0 : 1; }
      const int resident = (p.nblocks * gy + 255) / 256; p.prio_ranks = (norot || resident < 2) ? 0 : (resident > 4 ? 4 : resident);
      static int phase = -1; if (phase < 0) { const char* e = getenv("DA_PHASE_PRIO"); phase = (e && atoi(e)) ? 1 : 0; } if (phase) p.prio_ranks = -1; }
    p.s2in = S2dSrc{0, 0, 0, 0}; p.s2out = S2dSrc{0, 0, 0, 0};
    if (s2f && s2d_cin > 0) {
        if (s2f->fuse_in && !w_is_flipped_tr) p.s2in = S2dSrc{s2d_cin, s2f->D0, s2f->H0, s2f->W0};
        if (s2f->fuse_out && w_is_flipped_tr) p.s2out = S2dSrc{s2d_cin, s2f->D0, s2f->H0, s2f->W0};
    }
    p.stats_partial = stats_partial;
    if (stats_nparts) *stats_nparts = 0;
    p.ps1 = p.pt1 = p.ps2 = p.pt2 = nullptr; p.pslope1 = p.pslope2 = -1.f;
    p.dyn_ctr = dyn_ctr; p.wexp = dyn_ctr;
    if (pro) {
        const float *ones, *zeros;
        if (const int rc = pro_identity(&ones, &zeros)) return rc;
        if (pro_slopes(pro, C2, &p.pslope1, &p.pslope2)) return DA_ERR_UNSUPPORTED;
        p.ps1 = pro->s1 ? pro->s1 : ones; p.pt1 = pro->s1 ? pro->t1 : zeros;
        p.ps2 = (C2 > 0 && pro->s2) ? pro->s2 : ones; p.pt2 = (C2 > 0 && pro->s2) ? pro->t2 : zeros;
    }
    if (split) {
        if (stats_partial && stats_nparts) *stats_nparts = p.nblocks;
        // paired staging (one sector fetch per two chunks): one N-tile, an even number of 8-channel chunks that pair up inside in1 / in2
        static int nopair = -1; if (nopair < 0) { const char* e = getenv("DA_NO_PAIR"); nopair = (e && atoi(e)) ? 1 : 0; }
        static int wg3s = -1; if (wg3s < 0) { const char* e = getenv("DA_FWD_WG3"); wg3s = (e && atoi(e)) ? 1 : 0; }
        if (wg3s && NREP == 1 && !da_conv3_fwdsp_enabled())      // experiment: three workgroups per CU (<= 168 VGPRs), no paired staging
            return stats_partial ? (pro ? launch_fwd_mfma<8, 1, false, true, true, true, false, true, 0, false, false, 3>(p, gy, st) : launch_fwd_mfma<8, 1, false, true, true, false, false, true, 0, false, false, 3>(p, gy, st))
                                 : (pro ? launch_fwd_mfma<8, 1, false, false, true, true, false, true, 0, false, false, 3>(p, gy, st) : launch_fwd_mfma<8, 1, false, false, true, false, false, true, 0, false, false, 3>(p, gy, st));
        p.bst_y = nullptr; p.bst_par = nullptr; p.bst_slope = -1.f;
        if (da_conv3_fwdsp_enabled()) {     // weights in LDS, staging loads a whole item ahead (conv3d_fwdsp.hip); DA_FWDSP=0: the kernels below
            const bool pair = !nopair && NREP == 1 && !pro && !want_bst && C1 % 16 == 0 && C2 % 16 == 0;
            if (want_bst) {
                if (!(NREP == 1 && !pro && Cs2 == 0 && gy <= 2)) return DA_ERR_UNSUPPORTED;
                p.bst_y = g_bst.y; p.bst_par = g_bst.par; p.bst_slope = g_bst.slope;
            }
            return da_conv3_fwdsp_launch(p, gy, NREP, want_bst ? 2 : (stats_partial ? 1 : 0), pro ? 1 : 0, pair ? 1 : 0, st);
        }
        if (want_bst) {                                      // (da_conv3d_k3_dgrad_bst checked the shape: one output tensor of <= 32 channels, no prologue)
            if (!(NREP == 1 && !pro && Cs2 == 0 && gy <= 2)) return DA_ERR_UNSUPPORTED;
            p.bst_y = g_bst.y; p.bst_par = g_bst.par; p.bst_slope = g_bst.slope;
            return launch_fwd_mfma<8, 1, false, 2, true, false, false, true>(p, gy, st);      // (unpaired staging: with the pair's second parked chunk the eight y quads of the epilogue spill)
        }
        if (!nopair && NREP == 1 && !pro && C1 % 16 == 0 && C2 % 16 == 0)
            return stats_partial ? launch_fwd_mfma<8, 1, false, true, true, false, false, true, 0, true>(p, gy, st)
                                 : launch_fwd_mfma<8, 1, false, false, true, false, false, true, 0, true>(p, gy, st);
#define DA_SP_CASE(nr) if (NREP == nr) return stats_partial ? (pro ? launch_fwd_mfma<8, nr, false, true, true, true, false, true>(p, gy, st) : launch_fwd_mfma<8, nr, false, true, true, false, false, true>(p, gy, st)) \
                                                              : (pro ? launch_fwd_mfma<8, nr, false, false, true, true, false, true>(p, gy, st) : launch_fwd_mfma<8, nr, false, false, true, false, false, true>(p, gy, st))
        DA_SP_CASE(1); DA_SP_CASE(2);
#undef DA_SP_CASE
        return DA_ERR_UNSUPPORTED;
    }
    if (stats_partial && p.maskmode == 0 && (CK == 16 || CK == 8) && NREP <= 2) {
        if (stats_nparts) *stats_nparts = p.nblocks;
#define DA_ST_CASE(ck, nr) if (CK == ck && NREP == nr) return pro ? (hb ? launch_fwd_mfma<ck, nr, false, true, true, true, false, false, 0, false, true>(p, gy, st) : bf ? launch_fwd_mfma<ck, nr, false, true, true, true>(p, gy, st) : launch_fwd_mfma<ck, nr, false, true, false, true>(p, gy, st)) \
                                                                  : (hb ? launch_fwd_mfma<ck, nr, false, true, true, false, false, false, 0, false, true>(p, gy, st) : bf ? launch_fwd_mfma<ck, nr, false, true, true>(p, gy, st) : launch_fwd_mfma<ck, nr, false, true>(p, gy, st))
        DA_ST_CASE(16, 1); DA_ST_CASE(16, 2); DA_ST_CASE(8, 1); DA_ST_CASE(8, 2);
#undef DA_ST_CASE
    }
    if (pro) {
#define DA_PRO_CASE(ck, nr) if (CK == ck && NREP == nr) return hb ? launch_fwd_mfma<ck, nr, false, false, true, true, false, false, 0, false, true>(p, gy, st) : bf ? launch_fwd_mfma<ck, nr, false, false, true, true>(p, gy, st) : launch_fwd_mfma<ck, nr, false, false, false, true>(p, gy, st)
        DA_PRO_CASE(16, 1); DA_PRO_CASE(16, 2); DA_PRO_CASE(8, 1); DA_PRO_CASE(8, 2);
#undef DA_PRO_CASE
        return DA_ERR_UNSUPPORTED;
    }
    if (p.maskmode != 0) {
        const int s2f = p.s2in.cin > 0 ? 1 : (p.s2out.cin > 0 ? 2 : 0);
#define DA_M_CASE(nr, f) if (NREP == nr && s2f == f) return hb ? launch_fwd_mfma<16, nr, true, false, true, false, false, false, f, false, true>(p, gy, st) : bf ? launch_fwd_mfma<16, nr, true, false, true, false, false, false, f>(p, gy, st) : launch_fwd_mfma<16, nr, true, false, false, false, false, false, f>(p, gy, st)
        DA_M_CASE(1, 0); DA_M_CASE(1, 1); DA_M_CASE(1, 2); DA_M_CASE(2, 0); DA_M_CASE(2, 1); DA_M_CASE(2, 2);
#undef DA_M_CASE
        return DA_ERR_UNSUPPORTED;
    }
    if (dyn) {       // experiment (DA_DYN_TILES=1): work-stealing tile walk for the plain fp32 forward / data-gradient kernels
#define DA_DYN_CASE(ck, nr) if (CK == ck && NREP == nr) return launch_fwd_mfma<ck, nr, false, false, false, false, true>(p, gy, st)
        DA_DYN_CASE(16, 1); DA_DYN_CASE(16, 2); DA_DYN_CASE(16, 3); DA_DYN_CASE(8, 1); DA_DYN_CASE(8, 2);
#undef DA_DYN_CASE
    }
#define DA_FWD_CASE(ck, nr) if (CK == ck && NREP == nr) return bf ? launch_fwd_mfma<ck, nr, false, false, true>(p, gy, st) : launch_fwd_mfma<ck, nr>(p, gy, st)
#define DA_FWD_CASE_HB(ck, nr) if (hb && CK == ck && NREP == nr) return launch_fwd_mfma<ck, nr, false, false, true, false, false, false, 0, false, true>(p, gy, st)
    DA_FWD_CASE_HB(16, 1); DA_FWD_CASE_HB(16, 2); DA_FWD_CASE_HB(8, 1); DA_FWD_CASE_HB(8, 2);      // (bf16 mode: at most two N-tiles)
#undef DA_FWD_CASE_HB
    DA_FWD_CASE(16, 1); DA_FWD_CASE(16, 2); DA_FWD_CASE(16, 3);          // pick_nrep never asks for more than 3 N-tiles
    DA_FWD_CASE(8, 1); DA_FWD_CASE(8, 2); DA_FWD_CASE(8, 3);
#undef DA_FWD_CASE
    return DA_ERR_UNSUPPORTED;
}

// matrix mode of the 3x3x3 convolutions: process-wide switch (like da_set_conv_direct); the setters return the previous setting.
//   0  fp32 operands on v_mfma_f32_16x16x4_f32 (an fmaf chain)
//   1  operands ROUNDED to bf16 (BASELINE configs[4]'s precision; not fp32-accurate)
//   2  fp32 operands scaled per tile and split into two fp16 terms, three partial products per multiply on the fp16 pipe, fp32 accumulate
static int g_matrix_mode = -1;          // -1: not decided yet (env DA_MATRIX_MODE=0|1|2, or the older DA_MATRIX_BF16=1, for tools)
int da_matrix_mode() {
    if (g_matrix_mode < 0) {
        const char* m = getenv("DA_MATRIX_MODE"); const char* e = getenv("DA_MATRIX_BF16");
        g_matrix_mode = m ? atoi(m) : ((e && atoi(e) != 0) ? 1 : 0);
        if (g_matrix_mode < 0 || g_matrix_mode > 2) g_matrix_mode = 0;
    }
    return g_matrix_mode;
}
bool da_matrix_bf16() { return da_matrix_mode() == 1; }
extern "C" int da_set_matrix_bf16(int on) { const int prev = da_matrix_mode() == 1 ? 1 : 0; g_matrix_mode = on ? 1 : 0; return prev; }
extern "C" int da_set_matrix_mode(int mode) { const int prev = da_matrix_mode(); if (mode < 0 || mode > 2) return -1; g_matrix_mode = mode; return prev; }

static bool smallcin_ok(int C1, int C2, int Cout, int stride) { return stride == 1 && C1 + C2 <= 4 && Cout <= 32; }   // + 32-bit offsets, checked at launch
static const int kScBlocks = 1024;

bool da_conv3_thin_supported(int C1, int C2, int Cout, int stride) {
    const int Cin = C1 + C2;
    if (stride != 1) return false;
    if (Cout <= 4 && Cin >= 8 && Cin <= 64 && C1 % 8 == 0 && C2 % 8 == 0) return true;     // few outputs (flow forward)
    if (Cin <= 4 && Cout >= 8 && Cout <= 32) return true;                        // few inputs (flow data gradient, first encoder)
    return false;
}

static thread_local int g_thin_pack = 1, g_thin_only = 0;      // kept packed weights (da_pp_lookup in da_conv3_thin_fwd): pack at all / stop after the pack
static const size_t kThinPackBytes = 65536;        // padded weights [27][CinP][CT]: <= 27.6 KB (Cin <= 64, CT = 4) / 13.8 KB (Cin <= 4, CT <= 32)

template <int CL, int CT, int VPT, int JR = CT, int CR = CL>
static int thin_launch(ThinP& p, const float* w_src, float* wq, hipStream_t st, int j0 = 0, int Cw = -1) {
    const int Cin = p.C1 + p.C2;
    const int CinP = (Cin + CL - 1) / CL * CL;
    if ((size_t)27 * CinP * CT * sizeof(float) > kThinPackBytes / 2) return DA_ERR_UNSUPPORTED;
    if (g_thin_pack) hipLaunchKernelGGL(thin_pack_kernel, dim3(da_grid(27 * CinP * CT, 256, 64)), dim3(256), 0, st, w_src, wq, Cin, CinP, p.Cout, CT, p.flip_tr, j0, Cw < 0 ? p.Cout : Cw);
    if (g_thin_only) return 0;
    p.w = wq;
    const size_t ldsb = ((size_t)(2 * VPT + 2) * HY * HX * CL + (CT <= 16 ? 0 : (size_t)27 * CinP * CT)) * sizeof(float);
    p.ntz = (p.D + 2 * VPT - 1) / (2 * VPT);
    p.ntiles = p.N * p.ntz * p.nty * p.ntx;
    hipLaunchKernelGGL((conv3_thin_kernel<CL, CT, VPT, JR, CR>), dim3(p.ntiles), dim3(256), ldsb, st, p);
    return 0;
}

template <int CL, int CR = CL>
static int thin_few_inputs(ThinP& p, const float* w_src, float* wq, hipStream_t st) {
    if (p.Cout <= 8) return thin_launch<CL, 8, 4, 8, CR>(p, w_src, wq, st);
    if (p.Cout <= 16) return thin_launch<CL, 16, 4, 16, CR>(p, w_src, wq, st);
    static int split = -1; if (split < 0) { const char* e = getenv("DA_NO_THIN_SPLIT"); split = (e && atoi(e)) ? 0 : 1; }
    if (split && p.Cs2 > 0 && p.Cs1 <= 16 && p.Cs2 <= 16 && p.Cs1 % 4 == 0 && p.Cs2 % 4 == 0) {
        // split output (data gradient of a concat conv, e.g. the flow conv's 3 -> 16 + 8): one launch per output tensor, each with few
        // enough outputs per thread for scalar-cache weights; the (tiny) input is simply read twice
        ThinP a = p, b = p;
        a.Cout = p.Cs1; a.Cs2 = 0; a.out2 = nullptr;
        b.Cout = p.Cs2; b.out1 = p.out2; b.Cs1 = p.Cs2; b.Cs2 = 0; b.out2 = nullptr;
        float* wq2 = wq + kThinPackBytes / 2 / sizeof(float);
        int rc = (a.Cout <= 8) ? thin_launch<CL, 8, 4, 8, CR>(a, w_src, wq, st, 0, p.Cout) : thin_launch<CL, 16, 4, 16, CR>(a, w_src, wq, st, 0, p.Cout);
        if (rc) return rc;
        return (b.Cout <= 8) ? thin_launch<CL, 8, 4, 8, CR>(b, w_src, wq2, st, p.Cs1, p.Cout) : thin_launch<CL, 16, 4, 16, CR>(b, w_src, wq2, st, p.Cs1, p.Cout);
    }
    if (p.Cout <= 24) return thin_launch<CL, 24, 4, 24, CR>(p, w_src, wq, st);
    if (p.Cout <= 32) return thin_launch<CL, 32, 2, 32, CR>(p, w_src, wq, st);
    return DA_ERR_UNSUPPORTED;
}

int da_conv3_thin_fwd(const float* in1, int C1, const float* in2, int C2, const float* w, int flip_tr, const float* bias,
                      float* out1, int Cs1, float* out2, int Cs2, int N, int D, int H, int W, int Cout, float slope,
                      void* ws, size_t ws_bytes, hipStream_t st, int in_bf16, int out_bf16) {
    if ((unsigned long long)D * H * W * (C1 > C2 ? C1 : C2) * 4ull >= 0xFFFFFFF0ull) return DA_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < kThinPackBytes) return DA_ERR_WS_SMALL;
    const int Cin_ = C1 + C2;
    if (!(Cout <= 4 || Cin_ == 1 || (Cin_ <= 4 && C2 > 0) || (Cin_ <= 4 && C2 == 0))) return DA_ERR_UNSUPPORTED;      // (the cases below; decided before the kept-pack lookup)
    // the padded weights: in the workspace, or in the caller's kept buffer (conv3d_internal.h: da_pp_lookup; fp32 tensors only -- the bf16 twins pack per call)
    const DaKeptPack kp = (in_bf16 || out_bf16) ? DaKeptPack{nullptr, 0, 0} : da_pp_lookup(w, kThinPackBytes, flip_tr ? DA_PP_THIN_FLIP : DA_PP_THIN);
    if (kp.only && !kp.buf) return 0;
    float* wq = kp.buf ? (float*)kp.buf : (float*)ws;
    struct ThinFlags { ThinFlags(int pack, int only) { g_thin_pack = pack; g_thin_only = only; } ~ThinFlags() { g_thin_pack = 1; g_thin_only = 0; } } thin_flags((!kp.buf || kp.fill) ? 1 : 0, kp.only);
    ThinP p;
    p.in1 = in1; p.in2 = in2; p.C1 = C1; p.C2 = C2; p.w = w; p.bias = bias; p.out1 = out1; p.out2 = out2; p.Cs1 = Cs1; p.Cs2 = Cs2;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cout = Cout; p.flip_tr = flip_tr; p.slope = slope;
    p.nty = (H + TY - 1) / TY; p.ntx = (W + TX - 1) / TX;
    p.in_bf = in_bf16; p.out_bf = out_bf16;
    const int Cin = C1 + C2;
    int rc = 0;
    if (Cout <= 4) rc = thin_launch<8, 4, 2>(p, w, wq, st);      // (skipping the padded fourth output, JR = 3, was measured 1.7x SLOWER: 0.38 -> 0.63 ms)
    else if (Cin == 1 || (Cin <= 4 && C2 > 0)) rc = thin_few_inputs<1>(p, w, wq, st);      // CL = 1 splits cleanly at the concat boundary
    else if (Cin == 2 && (C2 == 0)) rc = thin_few_inputs<2>(p, w, wq, st);
    else if (Cin == 3 && C2 == 0) rc = thin_few_inputs<4, 3>(p, w, wq, st);
    else if (Cin <= 4 && C2 == 0) rc = thin_few_inputs<4>(p, w, wq, st);
    else return DA_ERR_UNSUPPORTED;
    if (rc) return rc;
    DA_LAUNCH_CHECK();
    return 0;
}

bool da_conv3_mfma_wgrad_supported(int C1, int C2, int Cout, int stride) {
    if (smallcin_ok(C1, C2, Cout, stride)) return true;
    if (stride != 1) return false;
    if (pick_ck(C1, C2) == 0) return false;
    if (Cout % 4 != 0 && Cout > 16) return false;      // scalar dY staging covers one cout tile
    return true;
}

// tmp[(t' * Cout + co) * Cin + ci] (t' = mirrored tap) -> dw_tio[((26 - t') * Cin + ci) * Cout + co]
__global__ void swapped_wgrad_place_kernel(const float* __restrict__ tmp, float* __restrict__ dw, int Cin, int Cout) {
    const int O = 27 * Cin * Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < O; i += gridDim.x * blockDim.x) {
        const int ci = i % Cin; const int r = i / Cin; const int co = r % Cout; const int t = r / Cout;
        dw[((size_t)(26 - t) * Cin + ci) * Cout + co] = tmp[i];
    }
}

template <int CK, int NREP, bool YS = false, bool MASKED = false, bool BF = false, bool PRO = false, bool SP = false, bool HB = false>
static int launch_wgrad_mfma(const WgP& p, const WgPlan& q, hipStream_t st) {
    const size_t shm = (size_t)(4 * HY * HX * CK + 2 * TY * TX * NREP * 16) * (BF ? 2 : 4);
    auto kern = conv3_mfma_wgrad_kernel<CK, NREP, YS, MASKED, BF, PRO, SP, HB>;
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

template <bool PRO, int NPL = 2, bool HB = false>
static int launch_split_wgrad(const WgP& p, const WgPlan& q, hipStream_t st) {
    size_t shm = (size_t)((DA_WG_TZ + 2) * (HY * HX * 8 + 4 * DA_WG_ZPAD) + DA_WG_TZ * TY * TX * 16) * 2 * NPL + 32;      // + the waves' tile maxima (split mode)
    if (shm < (size_t)2 * 15 * 64 * sizeof(float4)) shm = (size_t)2 * 15 * 64 * sizeof(float4);      // the cross-wave reduction at the end reuses the tiles' LDS
    auto kern = conv3_split_wgrad_kernel<PRO, NPL, HB>;
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

template <bool PRO, int NPL = 2, bool HB = false>
static int launch_split_wgrad16(const WgP& p, const WgPlan& q, hipStream_t st) {
    size_t shm = (size_t)(2 * 4 * (HY * HX * 8 + 4 * DA_WG16_ZPAD) + 2 * TY * TX * 16) * 2 * NPL + 128;      // the tile's planes + the waves' maxima
    if (shm < (size_t)4 * 15 * 64 * sizeof(float4)) shm = (size_t)4 * 15 * 64 * sizeof(float4);      // the cross-wave reduction at the end reuses the tile's LDS
    auto kern = conv3_split_wgrad16_kernel<PRO, NPL, HB>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(q.nslabs, q.nchunks, q.ngroups), dim3(512), shm, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}

template <bool PRO>
static int launch_split_wgrad16r(const WgP& p, const WgPlan& q, hipStream_t st) {
    // six plane slots x two half images x two fp16 planes + two dY buffers of two planes + the waves' maxima (2 parities x 3 x 8 floats)
    size_t shm = (size_t)(2 * 2 * 6 * (HY * HX * 8 + 4 * DA_WG16_ZPAD)) * 2 + (size_t)(2 * 2 * 2 * TY * TX * 16) * 2 + 2 * 24 * sizeof(float) + 8 * sizeof(float4);      // (+ the prologue's scale / shift quads)
    if (shm < (size_t)4 * 15 * 64 * sizeof(float4)) shm = (size_t)4 * 15 * 64 * sizeof(float4);
    auto kern = conv3_split_wgrad16r_kernel<PRO>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(q.nslabs, q.nchunks, q.ngroups), dim3(512), shm, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}

template <bool PRO>
static int launch_bf16_wgrad16r(const WgP& p, const WgPlan& q, hipStream_t st) {
    size_t shm = (size_t)(2 * 6 * (HY * HX * 8 + 4 * DA_WG16_ZPAD)) * 2 + (size_t)(2 * 2 * TY * TX * 16) * 2 + 8 * sizeof(float4);
    if (shm < (size_t)4 * 15 * 64 * sizeof(float4)) shm = (size_t)4 * 15 * 64 * sizeof(float4);
    auto kern = conv3_bf16_wgrad16r_kernel<PRO>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(q.nslabs, q.nchunks, q.ngroups), dim3(512), shm, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}

int da_conv3_mfma_wgrad(const float* in1, int C1, const float* in2, int C2, const float* dy, float* dw_tio,
                        int N, int D, int H, int W, int Cout, int stride, void* ws, size_t ws_bytes, hipStream_t st, int s2d_cin, const DaPro* pro, const DaS2dFuse* s2f, int act_bf16) {
    const bool hb = act_bf16 != 0;
    // bf16 activation storage: the matrix-core kernels of the bf16 matrix mode only (few-channel layers: the caller converts)
    if (hb && (da_matrix_mode() != 1 || pick_ck(C1, C2) == 0 || Cout % 4 != 0 || Cout <= 4 || smallcin_ok(C1, C2, Cout, stride) ||
               da_conv3_fewcin_wgrad_supported(C1, C2, Cout, stride))) return DA_ERR_UNSUPPORTED;
    if (pro && (stride != 1 || s2d_cin > 0 || C1 + C2 > kProMaxC || pick_ck(C1, C2) == 0 || Cout % 4 != 0 || Cout <= 4)) return DA_ERR_UNSUPPORTED;
    if (!pro && s2d_cin == 0 && da_conv3_fewcin_wgrad_supported(C1, C2, Cout, stride)) {
        static int off = -1; if (off < 0) { const char* e = getenv("DA_NO_FLOW_WGRAD"); off = (e && atoi(e)) ? 1 : 0; }
        if (!off) {
            const int rc = da_conv3_fewcin_wgrad(in1, C1, in2, C2, dy, dw_tio, N, D, H, W, Cout, ws, ws_bytes, st);
            if (rc != DA_ERR_UNSUPPORTED && rc != DA_ERR_WS_SMALL) return rc;
        }
    }
    if (smallcin_ok(C1, C2, Cout, stride)) {
        const int Cin = C1 + C2, O = 27 * Cin * Cout;
        if (ws_bytes < (size_t)kScBlocks * O * sizeof(float)) return DA_ERR_WS_SMALL;
        if ((unsigned long long)D * H * W * (Cout > Cin ? Cout : Cin) * 4ull >= 0xFFFFFFF0ull) return DA_ERR_UNSUPPORTED;
        ScP sp;
        sp.in1 = in1; sp.in2 = in2; sp.C1 = C1; sp.C2 = C2; sp.dy = dy; sp.partial = (float*)ws;
        sp.N = N; sp.D = D; sp.H = H; sp.W = W; sp.Cout = Cout; sp.nrows = (long long)N * D * H; sp.dyb = nullptr; sp.Cd1 = Cout;
        int nb = (int)da_cdiv(sp.nrows, 4); if (nb > kScBlocks) nb = kScBlocks;
        const int MT = (27 * Cin + 15) / 16, NT = (Cout + 15) / 16;
#define DA_SC_CASE(mt, nt) if (MT == mt && NT == nt) hipLaunchKernelGGL((conv3_smallcin_wgrad_kernel<mt, nt>), dim3(nb), dim3(256), 0, st, sp)
        DA_SC_CASE(2, 1); else DA_SC_CASE(2, 2); else DA_SC_CASE(4, 1); else DA_SC_CASE(4, 2); else DA_SC_CASE(6, 1); else DA_SC_CASE(6, 2);
        else DA_SC_CASE(7, 1); else DA_SC_CASE(7, 2); else return DA_ERR_UNSUPPORTED;
#undef DA_SC_CASE
        DA_LAUNCH_CHECK();
        { const int rc2 = da_reduce_partials(sp.partial, nb, O, dw_tio, st); if (rc2) return rc2; }
        return 0;
    }
    if (s2d_cin == 0 && !pro && da_conv3_flow_wgrad_supported(C1, C2, Cout, stride) && ws_bytes >= da_conv3_flow_wgrad_ws_bytes(C1 + C2, Cout)) {
        static int off = -1; if (off < 0) { const char* e = getenv("DA_NO_FLOW_WGRAD"); off = (e && atoi(e)) ? 1 : 0; }
        if (!off) {
            const int rc = da_conv3_flow_wgrad(in1, C1, in2, C2, dy, dw_tio, N, D, H, W, Cout, ws, ws_bytes, st);
            if (rc != DA_ERR_UNSUPPORTED) return rc;
        }
    }
    if (stride == 1 && s2d_cin == 0 && Cout <= 4 && C1 + C2 <= 32 && (unsigned long long)D * H * W * (C1 > C2 ? C1 : C2) * 4ull < 0xFFFFFFF0ull) {
        // Very few OUTPUT channels (the 24 -> 3 flow conv, voxel_morph.py:57): swap the operands.  dW[tap][ci][co] =
        // sum_u dy[u - tap][co] x[u][ci] is the small-Cin weight gradient of a conv with "input" dy (Cout channels), "output
        // gradient" x (Cin channels, possibly two tensors) and the taps mirrored; 27*Cout <= 108 rows instead of an MFMA N-tile
        // that is 13/16 padding.  The result [27][Cout][Cin] (mirrored) is permuted into dw_tio by a tiny kernel.
        const int Cin = C1 + C2, O = 27 * Cin * Cout;
        if (ws_bytes < (size_t)(kScBlocks + 1) * O * sizeof(float)) return DA_ERR_WS_SMALL;
        ScP sp;
        sp.in1 = dy; sp.in2 = nullptr; sp.C1 = Cout; sp.C2 = 0; sp.dy = in1; sp.dyb = in2; sp.Cd1 = C1; sp.Cout = Cin;
        sp.partial = (float*)ws; sp.N = N; sp.D = D; sp.H = H; sp.W = W; sp.nrows = (long long)N * D * H;
        int nb = (int)da_cdiv(sp.nrows, 4); if (nb > kScBlocks) nb = kScBlocks;
        const int MT = (27 * Cout + 15) / 16, NT = (Cin + 15) / 16;
#define DA_SC_CASE(mt, nt) if (MT == mt && NT == nt) hipLaunchKernelGGL((conv3_smallcin_wgrad_kernel<mt, nt>), dim3(nb), dim3(256), 0, st, sp)
        DA_SC_CASE(2, 1); else DA_SC_CASE(2, 2); else DA_SC_CASE(4, 1); else DA_SC_CASE(4, 2); else DA_SC_CASE(6, 1); else DA_SC_CASE(6, 2);
        else DA_SC_CASE(7, 1); else DA_SC_CASE(7, 2); else return DA_ERR_UNSUPPORTED;
#undef DA_SC_CASE
        DA_LAUNCH_CHECK();
        float* tmp = (float*)ws + (size_t)kScBlocks * O;
        { const int rc2 = da_reduce_partials(sp.partial, nb, O, tmp, st); if (rc2) return rc2; }
        hipLaunchKernelGGL(swapped_wgrad_place_kernel, dim3(da_grid(O, 256, 64)), dim3(256), 0, st, tmp, dw_tio, Cin, Cout);
        DA_LAUNCH_CHECK();
        return 0;
    }
    const bool split = da_matrix_mode() == 2 && s2d_cin == 0 && Cout % 4 == 0 && pick_ck(C1, C2) != 0;
    // bf16 matrix mode: the row-owner kernel with one operand plane (K = 32 MFMAs, a sixth of the split mode's matrix work); DA_BF16_WGRAD_V1=1
    // keeps the older tap-owner kernel (v_mfma_f32_16x16x16_bf16) for A/B
    static int bfv1 = -1; if (bfv1 < 0) { const char* e = getenv("DA_BF16_WGRAD_V1"); bfv1 = (e && atoi(e)) ? 1 : 0; }
    // (measured, 2 x 160 x 192 x 160: bf16 storage 48 -> 16 1.23 -> 1.09 ms, 16 -> 16 0.42 -> 0.38; NOT for more than one cout tile -- 96 -> 32: 0.44 ->
    // 0.70 ms, the kernel re-stages x per 16-cout group -- and not with fp32 tensors, 1.21 -> 1.94 ms: there the staging conversions dominate)
    const bool rows1 = !bfv1 && hb && da_matrix_mode() == 1 && s2d_cin == 0 && Cout % 4 == 0 && Cout <= 16 && pick_ck(C1, C2) != 0;
    const WgPlan q = wgrad_plan(N, D, H, W, C1, C2, Cout, split || rows1, split || rows1, (split && !hb) || rows1);
    if (!q.CK) return DA_ERR_UNSUPPORTED;
    const bool bf = da_matrix_bf16();      // (Cout % 4 != 0 keeps the exact kernel: its dY staging is scalar)
    if ((unsigned long long)D * H * W * 4ull * (unsigned long long)((C1 > C2 ? C1 : C2) > Cout ? (C1 > C2 ? C1 : C2) : Cout) >= 0xFFFFFFF0ull) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < q.partial_bytes + ((split || rows1) ? wg_tile_table_bytes(N, D, H, W) : 0)) return DA_ERR_WS_SMALL;
    WgP p;
    p.tiles = nullptr;
    p.s2in = (s2f && s2d_cin > 0 && s2f->fuse_in) ? S2dSrc{s2d_cin, s2f->D0, s2f->H0, s2f->W0} : S2dSrc{0, 0, 0, 0};
    { static int norot = -1; if (norot < 0) { const char* e = getenv("DA_PRIO_ROT"); norot = (e && atoi(e)) ? 0 : 1; }
      const int resident = (q.nslabs * q.nchunks * q.ngroups + 255) / 256; p.prio_ranks = (norot || resident < 2) ? 0 : (resident > 4 ? 4 : resident);
      static int phase = -1; if (phase < 0) { const char* e = getenv("DA_PHASE_PRIO"); phase = (e && atoi(e)) ? 1 : 0; } if (phase) p.prio_ranks = -1; }
    { static int abl = -1; if (abl < 0) { const char* e = getenv("DA_WG_ABLATE"); abl = e ? atoi(e) : 0; } p.ablate = abl; }
    if (split || rows1) {
        int4* tiles = reinterpret_cast<int4*>(reinterpret_cast<char*>(ws) + q.partial_bytes);
        if (q.w16 >= 2) tiles = nullptr;           // (the ring form derives its tiles from the position: no table, no launch)
        else hipLaunchKernelGGL(wgrad_tiles_kernel, dim3(da_grid(q.ntiles, 256, 256)), dim3(256), 0, st, tiles, q.ntiles, q.ntx, q.nty, q.ntz, DA_WG_TZ);
        DA_LAUNCH_CHECK();
        p.tiles = tiles;
    }
    p.in1 = in1; p.in2 = in2; p.C1 = C1; p.C2 = C2; p.dy = dy; p.partial = (float*)ws;
    p.N = N; p.D = D; p.H = H; p.W = W; p.Cout = Cout;
    p.ntz = q.ntz; p.nty = q.nty; p.ntx = q.ntx; p.ntiles = q.ntiles; p.tiles_per_slab = q.tps;
    p.O = 27 * (C1 + C2) * Cout;
    p.maskmode = 0;
    if (s2d_cin > 0) {
        if (q.CK != 16) return DA_ERR_UNSUPPORTED;
        p.maskmode = 1; da_s2d_masks(p.masks, q.nchunks, 16, s2d_cin, 0);
    }
    p.ps1 = p.pt1 = p.ps2 = p.pt2 = nullptr; p.pslope1 = p.pslope2 = -1.f;
    if (pro) {
        const float *ones, *zeros;
        if (const int rc0 = pro_identity(&ones, &zeros)) return rc0;
        if (pro_slopes(pro, C2, &p.pslope1, &p.pslope2)) return DA_ERR_UNSUPPORTED;
        p.ps1 = pro->s1 ? pro->s1 : ones; p.pt1 = pro->s1 ? pro->t1 : zeros;
        p.ps2 = (C2 > 0 && pro->s2) ? pro->s2 : ones; p.pt2 = (C2 > 0 && pro->s2) ? pro->t2 : zeros;
        int rcp = DA_ERR_UNSUPPORTED;
        if (split) rcp = q.w16 >= 2 ? launch_split_wgrad16r<true>(p, q, st) : q.w16 ? launch_split_wgrad16<true>(p, q, st) : launch_split_wgrad<true>(p, q, st);
        else if (rows1) rcp = q.w16 >= 2 ? launch_bf16_wgrad16r<true>(p, q, st) : q.w16 ? launch_split_wgrad16<true, 1, true>(p, q, st) : launch_split_wgrad<true, 1, true>(p, q, st);
        else
#define DA_WP_CASE(ck, nr) if (q.CK == ck && q.NREP == nr) rcp = hb ? launch_wgrad_mfma<ck, nr, false, false, true, true, false, true>(p, q, st) : bf ? launch_wgrad_mfma<ck, nr, false, false, true, true>(p, q, st) : launch_wgrad_mfma<ck, nr, false, false, false, true>(p, q, st)
        { DA_WP_CASE(16, 1); DA_WP_CASE(16, 2); DA_WP_CASE(8, 1); DA_WP_CASE(8, 2); }
#undef DA_WP_CASE
        if (rcp) return rcp;
        return da_reduce_partials(p.partial, q.nslabs, p.O, dw_tio, st);
    }
    int rc = DA_ERR_UNSUPPORTED;
    if (split) rc = q.w16 >= 2 ? launch_split_wgrad16r<false>(p, q, st) : q.w16 ? launch_split_wgrad16<false>(p, q, st) : launch_split_wgrad<false>(p, q, st);
    else if (rows1) rc = q.w16 >= 2 ? launch_bf16_wgrad16r<false>(p, q, st) : q.w16 ? launch_split_wgrad16<false, 1, true>(p, q, st) : launch_split_wgrad<false, 1, true>(p, q, st);
    else if (p.maskmode != 0) {
        if (Cout % 4 != 0) return DA_ERR_UNSUPPORTED;
        if (hb) rc = (q.NREP == 1) ? launch_wgrad_mfma<16, 1, false, true, true, false, false, true>(p, q, st) : launch_wgrad_mfma<16, 2, false, true, true, false, false, true>(p, q, st);
        else if (bf) rc = (q.NREP == 1) ? launch_wgrad_mfma<16, 1, false, true, true>(p, q, st) : launch_wgrad_mfma<16, 2, false, true, true>(p, q, st);
        else rc = (q.NREP == 1) ? launch_wgrad_mfma<16, 1, false, true>(p, q, st) : launch_wgrad_mfma<16, 2, false, true>(p, q, st);
    }
    else if (Cout % 4 != 0 && q.CK == 16) rc = launch_wgrad_mfma<16, 1, true>(p, q, st);
    else if (Cout % 4 != 0 && q.CK == 8) rc = launch_wgrad_mfma<8, 1, true>(p, q, st);
    else if (q.CK == 16 && q.NREP == 1) rc = hb ? launch_wgrad_mfma<16, 1, false, false, true, false, false, true>(p, q, st) : bf ? launch_wgrad_mfma<16, 1, false, false, true>(p, q, st) : launch_wgrad_mfma<16, 1>(p, q, st);
    else if (q.CK == 16 && q.NREP == 2) rc = hb ? launch_wgrad_mfma<16, 2, false, false, true, false, false, true>(p, q, st) : bf ? launch_wgrad_mfma<16, 2, false, false, true>(p, q, st) : launch_wgrad_mfma<16, 2>(p, q, st);
    else if (q.CK == 8 && q.NREP == 1) rc = hb ? launch_wgrad_mfma<8, 1, false, false, true, false, false, true>(p, q, st) : bf ? launch_wgrad_mfma<8, 1, false, false, true>(p, q, st) : launch_wgrad_mfma<8, 1>(p, q, st);
    else if (q.CK == 8 && q.NREP == 2) rc = hb ? launch_wgrad_mfma<8, 2, false, false, true, false, false, true>(p, q, st) : bf ? launch_wgrad_mfma<8, 2, false, false, true>(p, q, st) : launch_wgrad_mfma<8, 2>(p, q, st);
    if (rc) return rc;
    { const int rc2 = da_reduce_partials(p.partial, q.nslabs, p.O, dw_tio, st); if (rc2) return rc2; }
    return 0;
}

// ---- C ABI of the kept packs (see PrepackState) --------------------------------------------------------------------------------------------
extern "C" size_t da_conv3d_k3_pack_bytes(int N, int D, int H, int W, int Cin, int Cout) {
    if (Cin % 8 != 0 && Cout % 8 != 0) return 0;
    size_t a = Cin % 8 == 0 ? packed_bytes(Cin, Cout, 8) : 0;
    if (Cout % 8 == 0) { const size_t b = packed_bytes(Cout, Cin, 8); if (b > a) a = b; }      // (the data gradient: channels exchanged)
    return a + tile_table_bytes(N, D, H, W);
}
// Fill caller-owned buffer(s) with the packed operand of the stride-1 3x3x3 convolution C1 + C2 -> Cout on an N x D x H x W grid: the forward's
// (dgrad = 0) or the data gradient's (dgrad = 1; two regions when that call runs as two launches).  *used = regions filled; 0 = this shape / matrix
// mode does not run the split matrix-core kernels, keep nothing.  Only enqueues (one small kernel per region) on `stream`.
extern "C" int da_conv3d_k3_prepack(const float* w_tio, int C1, int C2, int Cout, int dgrad, int N, int D, int H, int W,
                                    void* b0, size_t n0, void* b1, size_t n1, int* used, void* stream) {
    if (used) *used = 0;
    if (!w_tio || !b0 || C1 <= 0 || C2 < 0 || Cout <= 0 || N <= 0) return DA_ERR_BADARG;
    if (da_matrix_mode() != 2) return 0;
    const int Cin = C1 + C2;
    if (!dgrad ? !da_conv3_mfma_fwd_supported(C1, C2, Cout, 1) : !da_conv3_mfma_fwd_supported(Cout, 0, Cin, 1, C1, C2)) return 0;
    g_pp = PrepackState{2, w_tio, {(char*)b0, (char*)b1}, {n0, b1 ? n1 : 0}, 0, 0};
    float* dummy = const_cast<float*>(w_tio);                   // (never dereferenced: in fill mode the call returns before its matrix kernel)
    const int rc = !dgrad ? da_conv3_mfma_fwd(dummy, C1, C2 > 0 ? dummy : nullptr, C2, w_tio, 0, nullptr, dummy, Cout, nullptr, 0, N, D, H, W, Cout, 1, -1.f,
                                              b0, 0, (hipStream_t)stream, 0, nullptr, nullptr, nullptr, nullptr, 0)
                          : da_conv3_mfma_fwd(dummy, Cout, nullptr, 0, w_tio, 1, nullptr, dummy, C1, C2 > 0 ? dummy : nullptr, C2, N, D, H, W, Cin, 1, -1.f,
                                              b0, 0, (hipStream_t)stream, 0, nullptr, nullptr, nullptr, nullptr, 0);
    const int n = g_pp.used;
    g_pp.mode = 0;
    if (rc == DA_ERR_UNSUPPORTED) return 0;
    if (rc) return rc;
    if (used) *used = n;
    return 0;
}
// The next stride-1 3x3x3 forward / data-gradient call of this thread on `w_tio` reads its packed operand(s) from here instead of packing (one call only;
// a call on other weights, or one that takes another kernel family, drops the hand-over).
extern "C" void da_conv3d_k3_use_prepacked(const float* w_tio, const void* b0, size_t n0, const void* b1, size_t n1) {
    g_pp = PrepackState{(w_tio && b0) ? 1 : 0, w_tio, {(char*)const_cast<void*>(b0), (char*)const_cast<void*>(b1)}, {n0, b1 ? n1 : 0}, 0, 0};
}

// da_conv3d_k3_prepack for `n` layers with as few launches as the kernel-argument size allows (48 regions per launch): what the host side runs after an
// optimiser step.  Arrays of length n; used[i] as in da_conv3d_k3_prepack.
extern "C" int da_conv3d_k3_prepack_many(int n, const float* const* w_tio, const int* C1, const int* C2, const int* Cout, const int* dgrad,
                                         const int* N, const int* D, const int* H, const int* W,
                                         void* const* b0, const size_t* n0, void* const* b1, const size_t* n1, int* used, void* stream) {
    if (n <= 0) return 0;
    if (!w_tio || !C1 || !C2 || !Cout || !dgrad || !N || !D || !H || !W || !b0 || !n0 || !b1 || !n1 || !used) return DA_ERR_BADARG;
    static thread_local PackJobs jobs;
    auto flush = [&]() -> int {
        if (g_pp_ncollect == 0) return 0;
        int gx = 1;
        for (int k = 0; k < g_pp_ncollect; ++k) if (jobs.j[k].Cin / 8 > gx) gx = jobs.j[k].Cin / 8;
        hipLaunchKernelGGL(pack_split_weights_many_kernel, dim3(gx, 8, g_pp_ncollect), dim3(256), 0, (hipStream_t)stream, jobs);
        g_pp_ncollect = 0;
        DA_LAUNCH_CHECK();
        return 0;
    };
    g_pp_collect = &jobs; g_pp_ncollect = 0;
    int rc = 0;
    for (int i = 0; i < n && !rc; ++i) {
        if (g_pp_ncollect + 2 > kPackJobsMax) rc = flush();
        if (!rc) rc = da_conv3d_k3_prepack(w_tio[i], C1[i], C2[i], Cout[i], dgrad[i], N[i], D[i], H[i], W[i], b0[i], n0[i], b1[i], n1[i], &used[i], stream);
    }
    if (!rc) rc = flush();
    g_pp_collect = nullptr; g_pp_ncollect = 0;
    return rc;
}

// da_conv3d_k3_dgrad of a layer whose FIRST input was act(BN(y)) applied on the fly (y = the producer's raw conv output, `stats4` = its statistics
// rows [mean | rstd | scale | shift][C1], `slope` its activation): besides dx1 -- the gradient with respect to the ACTIVATED tensor, as always -- the
// epilogue accumulates the producer's BatchNorm-backward sums, bst[*bst_n][2][C1] doubles = (sum dz, sum dz (y - mean)), dz = dx act'(y scale + shift),
// for da_bn_act_bwd_dbias_pre: the stand-alone reduction pass over (dx1, y) is not needed.  Split matrix mode; one-input layers of <= 16 channels, and
// concat layers (dx2 / C2: the decoder's 48 <- 16 data gradient, C1 = 32 up-sampled channels first, unets.py:275) as two launches -- dx1 with one
// N-tile per workgroup and the epilogue, dx2 as da_conv3d_k3_dgrad does it.  Anything else returns DA_ERR_UNSUPPORTED and the caller runs
// da_conv3d_k3_dgrad + the usual BatchNorm backward.  autograd of unets.py:30-32.
extern "C" int da_conv3d_k3_dgrad_bst(const float* dy, const float* w_tio, float* dx1, int C1, float* dx2, int C2, int N, int D, int H, int W, int Cout,
                                      const float* y, const float* stats4, float slope, double* bst, int bst_cap, int* bst_n,
                                      void* ws, size_t ws_bytes, void* stream) {
    if (bst_n) *bst_n = 0;
    if (!dy || !w_tio || !dx1 || !y || !stats4 || !bst || !bst_n || C1 <= 0 || C2 < 0 || (C2 > 0 && !dx2) || N <= 0 || Cout <= 0) return DA_ERR_BADARG;
    if (da_matrix_mode() != 2 || bst_cap < 512) return DA_ERR_UNSUPPORTED;
    if (C2 == 0) {
        if (C1 > 32 || C1 % 4 != 0 || (C1 > 16 && C1 % 16 != 0) || !da_conv3_mfma_fwd_supported(Cout, 0, C1, 1, C1, 0)) return DA_ERR_UNSUPPORTED;      // (one or two N-tiles)
        if (ws_bytes < da_conv3_mfma_ws_bytes(N, D, H, W, C1, Cout, 1)) return DA_ERR_WS_SMALL;
        g_bst = BstState{y, stats4, slope};
        const int rc = da_conv3_mfma_fwd(dy, Cout, nullptr, 0, w_tio, 1, nullptr, dx1, C1, nullptr, 0, N, D, H, W, C1, 1, -1.f, ws, ws_bytes, (hipStream_t)stream, 0, bst, bst_n);
        g_bst = BstState{nullptr, nullptr, -1.f};
        return rc;
    }
    if (C1 != 32 || C2 % 16 != 0 || C2 > 16 || !da_conv3_mfma_fwd_supported(Cout, 0, C1 + C2, 1, C1, C2) || pick_ck(Cout, 0) == 0) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_conv3_mfma_ws_bytes(N, D, H, W, C1 + C2, Cout, 1)) return DA_ERR_WS_SMALL;
    struct PpReset { ~PpReset() { if (g_pp.mode == 1) g_pp.mode = 0; } } pp_reset;
    if (g_pp.mode && g_pp.w != w_tio) g_pp.mode = 0;
    g_bst = BstState{y, stats4, slope};
    int rc = conv3_mfma_fwd_impl(dy, Cout, nullptr, 0, w_tio, 1, nullptr, dx1, C1, nullptr, 0, N, D, H, W, C1, 1, -1.f, ws, ws_bytes, (hipStream_t)stream, 0, bst, bst_n,
                                 nullptr, nullptr, 0, C1 + C2, false);
    g_bst = BstState{nullptr, nullptr, -1.f};
    if (rc) { *bst_n = 0; return rc; }
    rc = conv3_mfma_fwd_impl(dy, Cout, nullptr, 0, w_tio, 1, nullptr, dx2, C2, nullptr, 0, N, D, H, W, C2, 1, -1.f, ws, ws_bytes, (hipStream_t)stream, 0, nullptr, nullptr,
                             nullptr, nullptr, C1, C1 + C2, false);
    if (rc) *bst_n = 0;
    return rc;
}
