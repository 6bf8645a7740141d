// Split-mode (two-term fp16, split_f16.h) forward / data-gradient 3x3x3 convolution with the packed WEIGHTS IN LDS.
//
// Same GEMM view, tile (4 x 8 x 16 output voxels, 6 x 10 x 18 halo, 8-channel chunks, two fp16 planes), scale bookkeeping, epilogue and statistics as
// the split variants of conv3_mfma_fwd_kernel (conv3d_mfma.hip), which this kernel replaces for fp32 tensors.  What is different is the vector-memory
// schedule.  vmcnt retires in order, so in the older kernel a weight fragment (a 1 KiB buffer load per K-step, L2-resident) requested behind one of the
// next item's staging loads (HBM latency, microseconds under load) could not be used before that load had returned: every K-step of the first five
// waited for a staging load issued one step earlier, and the last staging load had two K-steps (~1500 cycles) to land before the hand-over barrier.
// Here
//   * the chunk's packed weights (14 KiB per N-tile) are copied global -> LDS by LDS-DMA (global_load_lds_dwordx4 through inline asm: no registers, no
//     ds_write pass, invisible to hipcc's vmcnt bookkeeping -- which stays exact for the ordinary loads -- and retired by the counted wait below) and the
//     K loop reads its A fragments with ds_read_b128 (lgkmcnt: independent of anything in flight to HBM); a layer with <= 2 chunks keeps all its weights
//     resident and never reloads; one N-tile: two buffers, the next chunk's copy runs under the current item; two N-tiles: one buffer, copied between
//     the two hand-over barriers (beside the conversion of the tile);
//   * ALL staging loads of the next item are issued at the top of the current item -- a whole item (~3 - 6 us) to land, nothing queued behind them that
//     the K loop needs.
// L1 traffic per item drops from 34.5 KB (tile) + 4 x 14 KB (every wave its own copy of the weights) to 34.5 + 14 KB.
#include "common.h"
#include "conv3d_internal.h"
#include "conv3d_stage.h"
#include <stdlib.h>
#include <stdio.h>

namespace {

constexpr int SP_CK = 8, SP_HZ = 6, SP_NSTEPS = 7;
#ifndef DA_FSP_BURST
#define DA_FSP_BURST 0      // 1: all staging loads of the next item at the top of the item (measured: 48 -> 16 forward 1.60 -> 2.15 ms with paired staging -- 18 loads
#endif                      // back to back, 32 cache lines each, hold the wave at vector-memory issue for ~2000 cycles; spread over the K-steps they issue between the MFMAs)
#ifndef DA_FSP_ABL
#define DA_FSP_ABL 0        // diagnostic builds only (wrong results): 8 no MFMAs, 16 no activation fragment reads after an item's first
#endif
#ifndef DA_FSP_TAIL
#define DA_FSP_TAIL 2       // K-steps at the end of an item without staging loads
#endif

// one 1 KiB piece global -> LDS: lane l copies 16 bytes from its `gsrc` to lds_dst + 16 l (lds_dst wave-uniform, in an SGPR).  M0 is written and
// restored inside the statement (the compiler does not preserve it around inline asm).
__device__ __forceinline__ void da_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NREP, int STATS, bool WG3 = false> struct SpLds {
    static constexpr int NWB = (NREP == 1 && !WG3) ? 2 : 1;              // weight buffers (WG3: three workgroups per CU, 50 KB each)
    static constexpr int WB = SP_NSTEPS * NREP * 2048;                   // bytes of one: [step][N-tile][plane h | l][lane][16 B]
    static constexpr int TILE_OFF = NWB * WB;                            // (the DMA targets sit below 64 KiB)
    static constexpr int TILE_B = StageGeom<SP_CK, SP_HZ>::TOTAL * 4 * 2 * 2;
    static constexpr int SRED_OFF = TILE_OFF + TILE_B;
    static constexpr int SRED_B = STATS ? 4 * 2 * NREP * 16 * 8 : 0;
    static constexpr int SMAX_OFF = SRED_OFF + SRED_B;
    static constexpr int TOTAL_B = SMAX_OFF + 16;
};

template <int NREP, int STATS, bool PRO, bool PAIR, bool WG3 = false>
__global__ void __launch_bounds__(256, WG3 ? 3 : 2) conv3_fwdsp_kernel(FwdP p) {
    static_assert(NREP == 1 || NREP == 2, "one or two N-tiles per workgroup");
    static_assert(STATS != 2 || (NREP == 1 && !PRO && !PAIR), "BatchNorm-backward sums: one N-tile, unpaired data gradient");
    static_assert(!PAIR || (NREP == 1 && !PRO), "paired staging: one N-tile, no input prologue");
    using L = SpLds<NREP, STATS, WG3>;
    using Frag = f16x8;
    constexpr int CK = SP_CK, HZ = SP_HZ, NSTEPS = SP_NSTEPS;
    constexpr int PRE = StageGeom<CK, HZ>::NIT;                        // 9 parked quads per thread and tile
    constexpr int PLANE_E = StageGeom<CK, HZ>::TOTAL * 4;             // two-byte elements per operand plane
    constexpr int NSTORES = TY * NREP;                                 // epilogue stores per lane and item (issued on every item, out of range unless last chunk)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* ldsb = reinterpret_cast<char*>(lds);
    float* tile = reinterpret_cast<float*>(ldsb + L::TILE_OFF);
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>(ldsb);    // LDS byte address of the dynamic segment (low half of the flat address)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 15, g = lane >> 4;
    const int nt0 = blockIdx.y * NREP;
    const int nchunks = (p.C1 + p.C2) / CK;
    const TileWalk tw = tile_walk(p.ntiles);
    const int nitems = tw.cnt * nchunks;
    if (nitems <= 0) {
        if (STATS) for (int c = threadIdx.x; c < NREP * 16; c += 256) { const int co = blockIdx.y * NREP * 16 + c; if (co < p.Cout) { p.stats_partial[((size_t)blockIdx.x * 2) * p.Cout + co] = 0.0; p.stats_partial[((size_t)blockIdx.x * 2 + 1) * p.Cout + co] = 0.0; } }
        return;
    }
    // weights: resident (every chunk has its own buffer for the whole launch) or re-copied per item
    const bool reload = nchunks > L::NWB;
    const char* wg = reinterpret_cast<const char*>(p.wp);
    auto copy_weights = [&](int chn, int buf) {                         // this wave's share of the 14 * NREP pieces of chunk chn -> weight buffer buf
#pragma unroll
        for (int k0 = 0; k0 < 14 * NREP; k0 += 4) {
            const int k = k0 + wave;
            if (k < 14 * NREP) {
                const int st = k / (2 * NREP), r = k - st * (2 * NREP);
                const char* src = wg + ((size_t)((chn * NSTEPS + st) * p.NT + nt0) * 2 + r) * 1024 + lane * 16;
                da_glds16(src, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * L::WB + k * 1024)));
            }
        }
    };

    // work item = (k-th tile of this workgroup's walk, channel chunk): the current and the next item's coordinates are kept in SGPRs
    int cK = 0, cCh = 0, cN, cZ, cY, cX, nK, nCh, nN, nZ, nY, nX;
    auto tile_at = [&](int k, int& n, int& z0, int& y0, int& x0) {
        int pos = tw.lo + k * tw.J; pos = pos < p.ntiles ? pos : p.ntiles - 1;          // (past the walk: any valid entry; never used)
        const int4 t = p.tiles[__builtin_amdgcn_readfirstlane(pos)];
        n = t.x; z0 = t.y; y0 = t.z; x0 = t.w;
    };
    auto advance = [&]() { nCh = cCh + 1; nK = cK; if (nCh == nchunks) { nCh = 0; nK = cK + 1; } tile_at(nK, nN, nZ, nY, nX); };
    tile_at(0, cN, cZ, cY, cX); advance();

    f32x4 acc[TY][NREP];
#pragma unroll
    for (int r = 0; r < TY; ++r)
#pragma unroll
        for (int nn = 0; nn < NREP; ++nn) acc[r][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float st1[STATS ? NREP : 1][4], st2[STATS ? NREP : 1][4];
#pragma unroll
    for (int nn = 0; nn < (STATS ? NREP : 1); ++nn)
#pragma unroll
        for (int j = 0; j < 4; ++j) { st1[nn][j] = 0.f; st2[nn][j] = 0.f; }
    // statistics: per-lane fp32 sums folded every 2 tiles into per-channel DOUBLE accumulators held by the first NREP * 16 threads (conv3d_mfma.hip)
    double dsum1 = 0.0, dsum2 = 0.0;
    double* sred = reinterpret_cast<double*>(ldsb + L::SRED_OFF);
    int tiles_done = 0;
    auto stats_flush = [&]() {
        if constexpr (STATS != 0) {
#pragma unroll
            for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double a = (double)st1[nn][j], b = (double)st2[nn][j];
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

    float4 pre[PRE];
    float4 pre2[PAIR ? PRE : 1];          // PAIR: the second chunk of the pair, loaded during the previous odd item
    unsigned vm = 0;
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psf = make_float4(0.f, 0.f, 0.f, 0.f); float pslope = -1.f;
    auto load_pro = [&](int ch) {
        if constexpr (PRO) {
            const int cbase = ch * CK;
            const bool first = cbase < p.C1;
            const int cofs = (first ? cbase : cbase - p.C1) + ((int)threadIdx.x % StageGeom<CK, HZ>::Q) * 4;
            psc = *reinterpret_cast<const float4*>((first ? p.ps1 : p.ps2) + cofs);
            psf = *reinterpret_cast<const float4*>((first ? p.pt1 : p.pt2) + cofs);
            pslope = first ? p.pslope1 : p.pslope2;
        }
    };
    // scale bookkeeping (wave-uniform): see conv3d_mfma.hip
    float* smax = reinterpret_cast<float*>(ldsb + L::SMAX_OFF);
    int Ecur = 0, Eacc = 0, Emin = 0, Enext = 0;
    auto sp_publish = [&](const float4* q) {
        const float m = da_wave_max_nonneg(stage_absmax<PRE>(q));
        if (lane == 0) smax[wave] = m;
    };
    auto sp_scale = [&](int chn) -> float {
        const float4 mm = *reinterpret_cast<const float4*>(smax);
        const int mi = __builtin_amdgcn_readfirstlane(__float_as_int(fmaxf(fmaxf(mm.x, mm.y), fmaxf(mm.z, mm.w))));
        const int ew = p.wexp[chn];
        int E = da_scale_exp(__int_as_float(mi)) + ew;
        if (chn != 0) E = min(E, Emin + 40);
        if (chn != 0 && E >= Ecur && E <= Ecur + 3) E = Ecur;
        Emin = (chn == 0) ? E : min(Emin, E);
        Enext = E;
        return da_pow2(E - ew);
    };
    StageMap<CK, HZ, true> smap; smap.init(p.H, p.W);
    // staging loads of one tile / chunk (and, PAIR, of the same voxels' next 8 channels into q2): load j of PRE
    __amdgpu_buffer_rsrc_t rsn;
    typename StageMap<CK, HZ, true>::Tile stile;
    auto loads_begin = [&](int n2, int z2, int y2, int x2, int ch2, bool valid) {
        const int cbase = ch2 * CK;
        const bool first = cbase < p.C1;
        const int Csn = first ? p.C1 : p.C2;
        const long long sample = (long long)p.D * p.H * p.W * Csn;
        rsn = da_rsrc_n<false>(first ? p.in1 : p.in2, n2, sample);
        stile = smap.tile(z2, y2, x2, p.D, p.H, p.W, Csn, first ? cbase : cbase - p.C1, valid && !(p.ablate & 1), 4);
        if constexpr (PRO) vm = 0;
    };
    auto load_one = [&](int j, auto Q2C) {      // (pre / pre2 are named directly: handed over as pointers they end up in scratch memory)
        const unsigned so = smap.offset(stile, j);
        pre[j] = da_buf_load4(rsn, so);
        if constexpr (decltype(Q2C)::value) pre2[j] = da_buf_load4(rsn, so == 0xFFFFFFFFu ? so : so + CK * 4u);
        if constexpr (PRO) vm |= (so != 0xFFFFFFFFu ? 1u : 0u) << j;
    };

    // ---- first item
    int wcur = 0;                                                      // weight buffer of the current item
    copy_weights(0, 0);
    if (!reload && nchunks == 2) copy_weights(1, 1);
    loads_begin(cN, cZ, cY, cX, 0, true);
#pragma unroll
    for (int j = 0; j < PRE; ++j) load_one(j, BoolC<PAIR>{});
    if constexpr (PRO) { load_pro(0); stage_pro_apply<0, PRE>(pre, vm, psc, psf, pslope); }
    sp_publish(pre);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the weight copies
    __syncthreads();
    { const float s0 = sp_scale(0); Ecur = Eacc = Enext; stage_write<CK, HZ, 0, PRE, true, true>(tile, pre, s0); }
    __syncthreads();

    const short* abase = reinterpret_cast<const short*>(tile) + ((wave * HY) * HX + i) * CK;
    int aoff32[NSTEPS];
#pragma unroll
    for (int t = 0; t < NSTEPS; ++t) { int tap = 4 * t + g; if (tap > 26) tap = 26; aoff32[t] = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * CK; }
    auto mma = [&](f32x4 c, const Frag& a, const Frag& b) -> f32x4 { return __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c, 0, 0, 0); };      // weights as A, activations as B: D = [cout][voxel]
    const int a4 = g;
    float bvv[NREP][4];
#pragma unroll
    for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int co = (nt0 + nn) * 16 + 4 * a4 + j; bvv[nn][j] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f; }

    // one work item; PH: 0 unpaired, 1 even item of a pair (the next item's tile is already parked in pre2), 2 odd item (loads both chunks of the next pair)
    auto item_body = [&](int item, auto PHC) {
        constexpr int PH = decltype(PHC)::value;
        const int n = cN, z0 = cZ, y0 = cY, x0 = cX, ch = cCh;
        const bool has_next = PH == 1 ? true : (item + 1 < nitems);
        const bool last = (ch == nchunks - 1);
        if (Ecur != Eacc) {          // bring the running sums into this item's unit (exact: a power of two)
            if (ch != 0) {
                const float f = da_acc_factor(Ecur - Eacc);
#pragma unroll
                for (int r = 0; r < TY; ++r)
#pragma unroll
                    for (int nn = 0; nn < NREP; ++nn) acc[r][nn] = acc[r][nn] * f;
            }
            Eacc = Ecur;
        }
        // the next item's weights (two buffers) and ALL of its staging loads, before anything else of this item touches vector memory
        if (L::NWB == 2 && reload && has_next) copy_weights(nCh, wcur ^ 1);
        if constexpr (PH != 1) {
            loads_begin(nN, nZ, nY, nX, nCh, has_next);
            if constexpr (DA_FSP_BURST) {
#pragma unroll
                for (int j = 0; j < PRE; ++j) load_one(j, BoolC<PH == 2>{});
            }
        }
        if constexpr (PRO) load_pro(has_next ? nCh : ch);

        // ---- K loop: 7 steps of 4 taps x 8 channels; per step 4 row blocks of 2 rows x 3 products x NREP N-tiles
        {
            const char* wl = ldsb + wcur * L::WB + lane * 16;
            Frag bq[2][NREP][2];
#pragma unroll
            for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) bq[0][nn][pl] = *reinterpret_cast<const Frag*>(wl + (nn * 2 + pl) * 1024);
            constexpr int RPB = 2;
            Frag AC[2][RPB];
            {
                const short* ap = abase + aoff32[0];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int rr = 0; rr < RPB; ++rr) AC[pl][rr] = *reinterpret_cast<const Frag*>(ap + pl * PLANE_E + rr * (HX * CK));
            }
#pragma unroll
            for (int s = 0; s < NSTEPS; ++s) {
                if constexpr (PH != 1 && !DA_FSP_BURST) {      // the next item's staging loads, spread over the first NSTEPS - TAIL steps (back to back they fill the CU's vector-memory queue and the wave waits at issue)
#pragma unroll
                    for (int j = 0; j < PRE; ++j)
                        if (j * (NSTEPS - DA_FSP_TAIL) / PRE == s) load_one(j, BoolC<PH == 2>{});
                }
                if (s + 1 < NSTEPS) {
#pragma unroll
                    for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) bq[(s + 1) & 1][nn][pl] = *reinterpret_cast<const Frag*>(wl + (((s + 1) * NREP + nn) * 2 + pl) * 1024);
                }
                const short* ap = abase + aoff32[s];
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};          // products small terms first: (a, b) planes (h, l) (l, h) (h, h)
                const short* anx = (s + 1 < NSTEPS) ? abase + aoff32[s + 1 < NSTEPS ? s + 1 : s] : ap;
#pragma unroll
                for (int qd = 0; qd < TY / RPB; ++qd) {
                    Frag AN[2][RPB];
                    const short* src = (qd + 1 < TY / RPB) ? ap + (RPB * qd + RPB) * (HX * CK) : anx;
#if !(DA_FSP_ABL & 16)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr) AN[pl][rr] = *reinterpret_cast<const Frag*>(src + pl * PLANE_E + rr * (HX * CK));
#else
                    (void)src; (void)AN;
#endif
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                            for (int rr = 0; rr < RPB; ++rr) {
#if DA_FSP_ABL & 8       // diagnostic build: no matrix instructions (the fragments are still read)
                                asm volatile("" :: "v"(AC[PA[pr]][rr]), "v"(bq[s & 1][nn][PB[pr]]));
#else
                                acc[RPB * qd + rr][nn] = mma(acc[RPB * qd + rr][nn], AC[PA[pr]][rr], bq[s & 1][nn][PB[pr]]);
#endif
                            }
                    __builtin_amdgcn_sched_barrier(0);
#if !(DA_FSP_ABL & 16)   // diagnostic build with bit 16: every product uses the first row block's fragments (no further activation reads)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr) AC[pl][rr] = AN[pl][rr];
#endif
                }
            }
        }

        // ---- epilogue (arithmetic on a tile's last chunk only; the stores are issued on every item, out of range -- dropped -- otherwise)
        {
            const int z = z0 + wave;
            const int x = x0 + i;
            const bool do_ep = last && !(p.ablate & 2);
            const float inv1 = da_pow2(-(Ecur / 2)), inv2 = da_pow2(-(Ecur - Ecur / 2));
            float4 yq[STATS == 2 ? TY : 1], bsc = make_float4(0.f, 0.f, 0.f, 0.f), bsf = bsc, bmu = bsc;
            if constexpr (STATS == 2) {
                const int cq = nt0 * 16 + 4 * a4;
                const bool cokq = do_ep && (cq + 3 < p.Cout) && z < p.D && x < p.W;
                const __amdgpu_buffer_rsrc_t ry = da_rsrc_n<false>(p.bst_y, n, (long long)p.D * p.H * p.W * p.Cs1);
#pragma unroll
                for (int r = 0; r < TY; ++r) {
                    const unsigned off = ((unsigned)((z * p.H + (y0 + r)) * p.W + x) * (unsigned)p.Cs1 + (unsigned)cq) * 4u;
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
                        float t0 = acc[r][nn][0], t1 = acc[r][nn][1], t2 = acc[r][nn][2], t3 = acc[r][nn][3];
                        t0 = t0 * inv1 * inv2; t1 = t1 * inv1 * inv2; t2 = t2 * inv1 * inv2; t3 = t3 * inv1 * inv2;      // back to the true unit (two exact factors: |E| may exceed 127)
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
                        if constexpr (STATS != 0) __builtin_amdgcn_sched_barrier(0);
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
                const bool cok = do_ep && !(p.ablate & 32) && (cb + 4 * a4 + 3 < p.Cout) && z < p.D && x < p.W;
                const long long sample = (long long)p.D * p.H * p.W * Cd;
                const __amdgpu_buffer_rsrc_t ro = da_rsrc_n<false>(dbase, n, sample);
#pragma unroll
                for (int r = 0; r < TY; ++r) {
                    const unsigned off = ((unsigned)((z * p.H + (y0 + r)) * p.W + x) * (unsigned)Cd + (unsigned)cd) * 4u;
                    da_buf_store4(ro, (cok && y0 + r < p.H) ? off : 0xFFFFFFFFu, acc[r][nn]);
                }
            }
            if (last) {
#pragma unroll
                for (int nn = 0; nn < NREP; ++nn)
#pragma unroll
                    for (int r = 0; r < TY; ++r) acc[r][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (STATS && last && !(p.ablate & 64) && ((++tiles_done) & 1) == 0) stats_flush();

        // ---- hand-over: the next item's tile (and, one buffer, weights) into LDS
        if (has_next && !(p.ablate & 4)) {
            if constexpr (PRO) stage_pro_apply<0, PRE>(pre, vm, psc, psf, pslope);
            sp_publish(PH == 1 ? pre2 : pre);
            if (L::NWB == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSTORES) : "memory");      // the weight copy issued at the top of this item (everything but the epilogue stores)
            __syncthreads();                       // every wave is done reading this item's tile and weights
            if (L::NWB == 1 && reload) copy_weights(nCh, 0);
            const float sn = sp_scale(nCh);
            stage_write<CK, HZ, 0, PRE, true, true>(tile, PH == 1 ? pre2 : pre, sn);
            if (L::NWB == 1 && reload) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        wcur = (L::NWB == 2) ? (reload ? (wcur ^ 1) : nCh) : 0;
        cK = nK; cCh = nCh; cN = nN; cZ = nZ; cY = nY; cX = nX;
        Ecur = Enext;
        advance();
    };
    if constexpr (PAIR) {
#pragma unroll 1
        for (int item = 0; item < nitems; item += 2) {
            item_body(item, IntC<1>{});
            item_body(item + 1, IntC<2>{});
        }
    } else {
#pragma unroll 1
        for (int item = 0; item < nitems; ++item) item_body(item, IntC<0>{});
    }
    if constexpr (STATS != 0) {
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


// ---------------------------------------------------------------------------------------------------------------------------------------------------
// Eight-wave pipelined form (one N-tile): ONE 512-thread workgroup per CU, TWO tile buffers, ONE barrier per item.
//
// The four-wave kernel above runs an item as a chain -- K loop, epilogue stores, wait for the next tile's loads, publish its maximum, barrier, convert +
// write it, barrier -- and only the CU's second workgroup overlaps one workgroup's chain links with the other's matrix work.  In the training step every
// link costs its full time (tools/ab/ablate_step.sh, seg step 20.1 ms: without the staging loads -2.1 ms, without the output stores -2.2, without
// conversion + barriers -1.3, without the MFMAs -2.9): the links add up instead of hiding under each other, the matrix pipe is busy 0.29 - 0.45.
// Here a wave never leaves its K loop for long.  During item i (tile buffer i & 1):
//   T1  the tile of item i + 1, parked in registers since item i - 1 and already folded into a published maximum, is scaled, split and written into the
//       OTHER tile buffer, one quad per row block, between the MFMAs;
//   T2  the registers it frees take the loads of item i + 2 (a whole item to land), one per row block; the weights of item i + 1 go global -> LDS by DMA;
//   T3  after the K loop (and the tile's epilogue on its last chunk) the wave publishes the maximum of item i + 2's quads;
//   one barrier: everything of item i + 1 is in LDS, everyone is done with item i.
// Wave w owns z slab w & 3 and the row half w >> 2 (4 M-tiles): 84 MFMAs per item, two waves per SIMD.  Statistics: per-lane fp32 sums over two tiles, a
// DPP butterfly over the 16 voxel lanes in double, per-lane double accumulators -- no LDS strip, no extra barriers; one cross-wave reduction at the end.
// ---------------------------------------------------------------------------------------------------------------------------------------------------
struct Sp8Lds {
    static constexpr int WB = SP_NSTEPS * 2048;                          // one chunk's packed weights of one N-tile
    static constexpr int TILE_OFF = 2 * WB;
    static constexpr int TILE_B = StageGeom<SP_CK, SP_HZ>::TOTAL * 4 * 2 * 2;
    static constexpr int SMAX_OFF = TILE_OFF + 2 * TILE_B;               // [2 parities][8 waves] floats
    static constexpr int TOTAL_B = SMAX_OFF + 64;
};

// all-reduce over the 16 lanes of a DPP row (same g, every i): quad permutes, half-row mirror, row mirror -- VALU only
template <int CTRL> __device__ __forceinline__ double da_dpp_add_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double da_row16_sum(double v) {
    v = da_dpp_add_f64<0xB1>(v); v = da_dpp_add_f64<0x4E>(v); v = da_dpp_add_f64<0x141>(v); v = da_dpp_add_f64<0x140>(v);
    return v;
}

template <int STATS, bool PRO, bool PAIR>
__global__ void __launch_bounds__(512, 1) conv3_fwdsp8_kernel(FwdP p) {
    static_assert(STATS != 2 || (!PRO && !PAIR), "BatchNorm-backward sums: unpaired data gradient");
    static_assert(!PAIR || !PRO, "paired staging: no input prologue");
    using L = Sp8Lds;
    using Frag = f16x8;
    constexpr int CK = SP_CK, NSTEPS = SP_NSTEPS, NT = 512;
    constexpr int TOTQ = StageGeom<SP_CK, SP_HZ>::TOTAL;               // 2160 quads per tile and chunk
    constexpr int NIT = (TOTQ + NT - 1) / NT;                          // 5 parked quads per thread
    constexpr int PLANE_E = TOTQ * 4;
    constexpr int RW = 4;                                              // rows (M-tiles) per wave
    constexpr int NSTORES = RW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* ldsb = reinterpret_cast<char*>(lds);
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>(ldsb);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int zs = wave & 3, hh = wave >> 2;
    const int i = lane & 15, g = lane >> 4;
    const int nt0 = blockIdx.y;
    const int nchunks = (p.C1 + p.C2) / CK;
    const TileWalk tw = tile_walk(p.ntiles);
    const int nitems = tw.cnt * nchunks;
    // the caller reduces 2 * gridDim.x rows of partial statistics (it sized them for the four-wave grid): this workgroup's sums go to row blockIdx.x,
    // zeros to row gridDim.x + blockIdx.x
    auto stats_rows = [&](double a, double b) {
        if constexpr (STATS != 0) {
            if ((int)threadIdx.x < 16) {
                const int co = nt0 * 16 + (int)threadIdx.x;
                if (co < p.Cout) {
                    p.stats_partial[((size_t)blockIdx.x * 2 + 0) * p.Cout + co] = a;
                    p.stats_partial[((size_t)blockIdx.x * 2 + 1) * p.Cout + co] = b;
                    p.stats_partial[((size_t)(gridDim.x + blockIdx.x) * 2 + 0) * p.Cout + co] = 0.0;
                    p.stats_partial[((size_t)(gridDim.x + blockIdx.x) * 2 + 1) * p.Cout + co] = 0.0;
                }
            }
        }
    };
    if (nitems <= 0) { stats_rows(0.0, 0.0); return; }
    const bool reload = nchunks > 2;
    const char* wg = reinterpret_cast<const char*>(p.wp);
    auto copy_weights = [&](int chn, int buf) {
#pragma unroll
        for (int k0 = 0; k0 < 14; k0 += 8) {
            const int k = k0 + wave;
            if (k < 14) {
                const int st = k >> 1, r = k & 1;
                const char* src = wg + ((size_t)((chn * NSTEPS + st) * p.NT + nt0) * 2 + r) * 1024 + lane * 16;
                da_glds16(src, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * L::WB + k * 1024)));
            }
        }
    };
    // item coordinates: c = current, n = next (being converted), m = the one after (being loaded); SGPRs
    struct Item { int k, ch, n, z, y, x; };
    auto tile_at = [&](Item& it) {
        int pos = tw.lo + it.k * tw.J; pos = pos < p.ntiles ? pos : p.ntiles - 1;
        const int4 t = p.tiles[__builtin_amdgcn_readfirstlane(pos)];
        it.n = t.x; it.z = t.y; it.y = t.z; it.x = t.w;
    };
    auto succ = [&](const Item& a) -> Item { Item b; b.ch = a.ch + 1; b.k = a.k; if (b.ch == nchunks) { b.ch = 0; b.k = a.k + 1; } tile_at(b); return b; };
    Item cI; cI.k = 0; cI.ch = 0; tile_at(cI);
    Item nI = succ(cI), mI = succ(nI);

    f32x4 acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
    double da1[STATS ? 4 : 1], da2[STATS ? 4 : 1];
#pragma unroll
    for (int j = 0; j < (STATS ? 4 : 1); ++j) { da1[j] = 0.0; da2[j] = 0.0; }
    int tiles_done = 0;
    auto stats_fold = [&]() {                                           // per-lane fp32 sums -> double, summed over the row's 16 voxel lanes, into per-lane double accumulators
        if constexpr (STATS != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                da1[j] += da_row16_sum((double)st1[j]); da2[j] += da_row16_sum((double)st2[j]);
                st1[j] = 0.f; st2[j] = 0.f;
            }
        }
    };

    // staging map (launch constants): quad idx = threadIdx.x + 512 it -> halo voxel idx / 2, channel quad idx & 1
    unsigned pk[NIT]; int vo[NIT];
    const int c4x4 = ((int)threadIdx.x & 1) * 4;
    const bool small = (long long)SP_HZ * p.H * p.W < (1ll << 24);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = (int)threadIdx.x + it * NT, hv = idx >> 1;
        const int hx = hv % HX, t = hv / HX, hy = t % HY, hz = t / HY;
        pk[it] = idx < TOTQ ? ((unsigned)hz << 16 | (unsigned)hy << 8 | (unsigned)hx) : 0xFFFF0000u;
        vo[it] = idx < TOTQ ? (hz * p.H + hy) * p.W + hx : 0;
    }
    float4 pre[NIT];
    float4 pre2[PAIR ? NIT : 1];
    unsigned vm = 0;
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psf = make_float4(0.f, 0.f, 0.f, 0.f); float pslope = -1.f;
    auto load_pro = [&](int ch) {
        if constexpr (PRO) {
            const int cbase = ch * CK;
            const bool first = cbase < p.C1;
            const int cofs = (first ? cbase : cbase - p.C1) + c4x4;
            psc = *reinterpret_cast<const float4*>((first ? p.ps1 : p.ps2) + cofs);
            psf = *reinterpret_cast<const float4*>((first ? p.pt1 : p.pt2) + cofs);
            pslope = first ? p.pslope1 : p.pslope2;
        }
    };
    // loads of one item: tile-level part (wave-uniform), then quad j
    __amdgpu_buffer_rsrc_t rsn;
    bool t_int = false, t_valid = false; int t_z = 0, t_y = 0, t_x = 0, t_Cs4 = 0, t_cofs4 = 0; unsigned t_base = 0;
    auto loads_begin = [&](const Item& it, bool valid) {
        const int cbase = it.ch * CK;
        const bool first = cbase < p.C1;
        const int Csn = first ? p.C1 : p.C2;
        rsn = da_rsrc_n<false>(first ? p.in1 : p.in2, it.n, (long long)p.D * p.H * p.W * Csn);
        t_valid = valid && !(p.ablate & 1); t_z = it.z; t_y = it.y; t_x = it.x;
        t_int = small && t_valid && it.z >= 1 && it.z + SP_HZ - 2 < p.D && it.y >= 1 && it.y + HY - 2 < p.H && it.x >= 1 && it.x + HX - 2 < p.W;
        t_Cs4 = Csn * 4; t_cofs4 = ((first ? cbase : cbase - p.C1) + c4x4) * 4;
        t_base = (unsigned)(((it.z - 1) * p.H + (it.y - 1)) * p.W + (it.x - 1)) * (unsigned)t_Cs4 + (unsigned)t_cofs4;
        if constexpr (PRO) vm = 0;
    };
    auto load_off = [&](int j) -> unsigned {
        const int hz = (int)(pk[j] >> 16), hy = (int)((pk[j] >> 8) & 255u), hx = (int)(pk[j] & 255u);
        if (t_int) { const unsigned o = __umul24((unsigned)vo[j], (unsigned)t_Cs4) + t_base; return ((j + 1) * NT <= TOTQ || hz != 0xFFFF) ? o : 0xFFFFFFFFu; }
        const int z = t_z - 1 + hz, y = t_y - 1 + hy, x = t_x - 1 + hx;
        const bool inb = t_valid && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        return inb ? (unsigned)(((z * p.H + y) * p.W + x) * t_Cs4 + t_cofs4) : 0xFFFFFFFFu;
    };
    auto load_one = [&](int j, auto Q2C) {
        const unsigned so = load_off(j);
        pre[j] = da_buf_load4(rsn, so);
        if constexpr (decltype(Q2C)::value) pre2[j] = da_buf_load4(rsn, so == 0xFFFFFFFFu ? so : so + CK * 4u);
        if constexpr (PRO) vm |= (so != 0xFFFFFFFFu ? 1u : 0u) << j;
    };
    auto pro_apply = [&]() { if constexpr (PRO) stage_pro_apply<0, NIT>(pre, vm, psc, psf, pslope); };
    // scales
    float* smax = reinterpret_cast<float*>(ldsb + L::SMAX_OFF);
    int Ecur = 0, Eacc = 0, Emin = 0, Enext = 0;
    auto publish = [&](const float4* q, int par) {
        const float m = da_wave_max_nonneg(stage_absmax<NIT>(q));
        if (lane == 0) smax[par * 8 + wave] = m;
    };
    auto next_scale = [&](int par, int chn) -> float {                 // after the barrier: the scale of the tile about to be converted (chunk chn); sets Enext
        const float4 ma = *reinterpret_cast<const float4*>(smax + par * 8), mb = *reinterpret_cast<const float4*>(smax + par * 8 + 4);
        const float mx = fmaxf(fmaxf(fmaxf(ma.x, ma.y), fmaxf(ma.z, ma.w)), fmaxf(fmaxf(mb.x, mb.y), fmaxf(mb.z, mb.w)));
        const int mi = __builtin_amdgcn_readfirstlane(__float_as_int(mx));
        const int ew = p.wexp[chn];
        int E = da_scale_exp(__int_as_float(mi)) + ew;
        if (chn != 0) E = min(E, Emin + 40);
        if (chn != 0 && E >= Ecur && E <= Ecur + 3) E = Ecur;
        Emin = (chn == 0) ? E : min(Emin, E);
        Enext = E;
        return da_pow2(E - ew);
    };
    auto convert_one = [&](const float4 v, int j, float sc, int buf) {   // quad j of this thread -> both planes of tile buffer buf
        const int idx = (int)threadIdx.x + j * NT;
        if (idx < TOTQ) {
            uint2 h, l; da_split2(v, sc, h, l);
            uint2* t = reinterpret_cast<uint2*>(ldsb + L::TILE_OFF + buf * L::TILE_B);
            t[idx] = h; t[idx + TOTQ] = l;
        }
    };

    // ---- prologue: item 0 into tile buffer 0, item 1 parked with its maximum published
    copy_weights(0, 0);
    if (!reload && nchunks == 2) copy_weights(1, 1);
    loads_begin(cI, true);
#pragma unroll
    for (int j = 0; j < NIT; ++j) load_one(j, BoolC<PAIR>{});
    if constexpr (PRO) { load_pro(0); pro_apply(); }
    publish(pre, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const float s0 = next_scale(0, 0); Ecur = Eacc = Enext;
#pragma unroll
        for (int j = 0; j < NIT; ++j) convert_one(pre[j], j, s0, 0);
    }
    if constexpr (!PAIR) {
        loads_begin(nI, nitems > 1);
#pragma unroll
        for (int j = 0; j < NIT; ++j) load_one(j, BoolC<false>{});
        if constexpr (PRO) { load_pro(nI.ch); pro_apply(); }
        publish(pre, 1);
    } else publish(pre2, 1);
    int wcur = 0;

    int aoff32[NSTEPS];
#pragma unroll
    for (int t = 0; t < NSTEPS; ++t) { int tap = 4 * t + g; if (tap > 26) tap = 26; aoff32[t] = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * CK; }
    auto mma = [&](f32x4 c, const Frag& a, const Frag& b) -> f32x4 { return __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c, 0, 0, 0); };
    const int a4 = g;
    float bvv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int co = nt0 * 16 + 4 * a4 + j; bvv[j] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f; }
    const int laneA = ((zs * HY + RW * hh) * HX + i) * CK;               // this wave's first row in a tile buffer (two-byte elements)

    // PH: 0 unpaired; 1 even item of a pair (converts pre2, loads both chunks of the next pair); 2 odd item (converts pre, loads nothing)
    auto item_body = [&](int item, auto PHC) {
        constexpr int PH = decltype(PHC)::value;
        const int n = cI.n, z0 = cI.z, y0 = cI.y, x0 = cI.x, ch = cI.ch;
        const int bufc = item & 1, bufn = bufc ^ 1;
        const bool last = (ch == nchunks - 1);
        const bool has1 = item + 1 < nitems, has2 = item + 2 < nitems;
        __syncthreads();                                                 // item `item` is in LDS (tile + weights), its successor's maxima are published; everyone is done with item - 1
        if (Ecur != Eacc) {
            if (ch != 0) {
                const float f = da_acc_factor(Ecur - Eacc);
#pragma unroll
                for (int r = 0; r < RW; ++r) acc[r] = acc[r] * f;
            }
            Eacc = Ecur;
        }
        const float sn = next_scale(bufn, nI.ch);                        // (item + 1 past the walk: zeros at any scale)
        if constexpr (PH != 2) loads_begin(mI, has2);
        // side work of this item, one unit per row block u = 0 .. 13: T1 quads 0 .. 4, then T2 loads 0 .. 4, then the weight copy
        auto side = [&](int u) {
            if (!(p.ablate & 4)) { if (u < NIT) convert_one(PH == 1 ? pre2[PAIR ? u : 0] : pre[u], u, sn, bufn); }
            if constexpr (PH != 2) { if (u >= NIT && u < 2 * NIT) load_one(u - NIT, BoolC<PH == 1>{}); }
            if (u == 2 * NIT) { if (reload && has1) copy_weights(nI.ch, wcur ^ 1); if constexpr (PRO) load_pro(has2 ? mI.ch : ch); }
        };
        {
            const char* wl = ldsb + wcur * L::WB + lane * 16;
            const short* abase = reinterpret_cast<const short*>(ldsb + L::TILE_OFF + bufc * L::TILE_B) + laneA;
            Frag bq[2][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bq[0][pl] = *reinterpret_cast<const Frag*>(wl + pl * 1024);
            constexpr int RPB = 2;
            Frag AC[2][RPB];
            {
                const short* ap = abase + aoff32[0];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int rr = 0; rr < RPB; ++rr) AC[pl][rr] = *reinterpret_cast<const Frag*>(ap + pl * PLANE_E + rr * (HX * CK));
            }
#pragma unroll
            for (int s = 0; s < NSTEPS; ++s) {
                if (s + 1 < NSTEPS) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) bq[(s + 1) & 1][pl] = *reinterpret_cast<const Frag*>(wl + ((s + 1) * 2 + pl) * 1024);
                }
                const short* ap = abase + aoff32[s];
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
                const short* anx = abase + aoff32[s + 1 < NSTEPS ? s + 1 : s];
#pragma unroll
                for (int qd = 0; qd < RW / RPB; ++qd) {
                    Frag AN[2][RPB];
                    const short* src = (qd + 1 < RW / RPB) ? ap + (RPB * qd + RPB) * (HX * CK) : anx;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr) AN[pl][rr] = *reinterpret_cast<const Frag*>(src + pl * PLANE_E + rr * (HX * CK));
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr)
                            acc[RPB * qd + rr] = mma(acc[RPB * qd + rr], AC[PA[pr]][rr], bq[s & 1][PB[pr]]);
                    side(s * (RW / RPB) + qd);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr) AC[pl][rr] = AN[pl][rr];
                }
            }
        }
        // ---- epilogue of a tile's last chunk (stores issued on every item, out of range -- dropped -- otherwise)
        {
            const int z = z0 + zs;
            const int x = x0 + i;
            const int yb = y0 + RW * hh;
            const bool do_ep = last && !(p.ablate & 2);
            const float inv1 = da_pow2(-(Ecur / 2)), inv2 = da_pow2(-(Ecur - Ecur / 2));
            float4 yq[STATS == 2 ? RW : 1], bsc = make_float4(0.f, 0.f, 0.f, 0.f), bsf = bsc, bmu = bsc;
            if constexpr (STATS == 2) {
                const int cq = nt0 * 16 + 4 * a4;
                const bool cokq = do_ep && (cq + 3 < p.Cout) && z < p.D && x < p.W;
                const __amdgpu_buffer_rsrc_t ry = da_rsrc_n<false>(p.bst_y, n, (long long)p.D * p.H * p.W * p.Cs1);
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const unsigned off = ((unsigned)((z * p.H + (yb + r)) * p.W + x) * (unsigned)p.Cs1 + (unsigned)cq) * 4u;
                    yq[r] = da_buf_load4(ry, (cokq && yb + r < p.H) ? off : 0xFFFFFFFFu);
                }
                const __amdgpu_buffer_rsrc_t rp = da_rsrc(p.bst_par, (unsigned)(4 * p.Cs1 * 4));
                const unsigned po = (do_ep && cq + 3 < p.Cout) ? (unsigned)(cq * 4) : 0xFFFFFFFFu;
                bmu = da_buf_load4(rp, po);
                bsc = da_buf_load4(rp, po == 0xFFFFFFFFu ? po : po + (unsigned)(2 * p.Cs1 * 4));
                bsf = da_buf_load4(rp, po == 0xFFFFFFFFu ? po : po + (unsigned)(3 * p.Cs1 * 4));
            }
            if (do_ep) {
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    float t0 = acc[r][0], t1 = acc[r][1], t2 = acc[r][2], t3 = acc[r][3];
                    t0 = t0 * inv1 * inv2; t1 = t1 * inv1 * inv2; t2 = t2 * inv1 * inv2; t3 = t3 * inv1 * inv2;
                    const float v0 = t0 + bvv[0], v1 = t1 + bvv[1], v2 = t2 + bvv[2], v3 = t3 + bvv[3];
                    if constexpr (STATS == 2) {
                        const float m = (z < p.D && yb + r < p.H && x < p.W) ? 1.f : 0.f;
                        const float4 yv = yq[r];
                        const float d0 = m * v0 * da_act_grad(yv.x * bsc.x + bsf.x, p.bst_slope), d1 = m * v1 * da_act_grad(yv.y * bsc.y + bsf.y, p.bst_slope);
                        const float d2 = m * v2 * da_act_grad(yv.z * bsc.z + bsf.z, p.bst_slope), d3 = m * v3 * da_act_grad(yv.w * bsc.w + bsf.w, p.bst_slope);
                        st1[0] += d0; st1[1] += d1; st1[2] += d2; st1[3] += d3;
                        st2[0] += d0 * (yv.x - bmu.x); st2[1] += d1 * (yv.y - bmu.y); st2[2] += d2 * (yv.z - bmu.z); st2[3] += d3 * (yv.w - bmu.w);
                    } else if (STATS) {
                        const float m = (z < p.D && yb + r < p.H && x < p.W) ? 1.f : 0.f;
                        st1[0] += m * v0; st1[1] += m * v1; st1[2] += m * v2; st1[3] += m * v3;
                        st2[0] += m * v0 * v0; st2[1] += m * v1 * v1; st2[2] += m * v2 * v2; st2[3] += m * v3 * v3;
                    }
                    acc[r] = (f32x4){da_act(v0, p.slope), da_act(v1, p.slope), da_act(v2, p.slope), da_act(v3, p.slope)};
                }
            }
            {
                const int cb = nt0 * 16;
                const bool first = cb < p.Cs1;
                float* dbase = first ? p.out1 : p.out2;
                const int Cd = first ? p.Cs1 : p.Cs2;
                const int cd = cb - (first ? 0 : p.Cs1) + 4 * a4;
                const bool cok = do_ep && !(p.ablate & 32) && (cb + 4 * a4 + 3 < p.Cout) && z < p.D && x < p.W;
                const __amdgpu_buffer_rsrc_t ro = da_rsrc_n<false>(dbase, n, (long long)p.D * p.H * p.W * Cd);
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const unsigned off = ((unsigned)((z * p.H + (yb + r)) * p.W + x) * (unsigned)Cd + (unsigned)cd) * 4u;
                    da_buf_store4(ro, (cok && yb + r < p.H) ? off : 0xFFFFFFFFu, acc[r]);
                }
            }
            if (last) {
#pragma unroll
                for (int r = 0; r < RW; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (STATS && last && ((++tiles_done) & 3) == 0) stats_fold();      // (4 rows per tile and lane: <= 16 values per fp32 sum, as in the four-wave kernel)
        // ---- T3: the maximum of item + 2's quads (loaded during this item; PAIR, odd item: parked since the previous one)
        if constexpr (PH == 2) publish(pre2, bufc);
        else { pro_apply(); publish(pre, bufc); }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSTORES) : "memory");     // the weight copy of this item (everything older than the epilogue stores)
        wcur = reload ? (wcur ^ 1) : nI.ch;
        cI = nI; nI = mI; mI = succ(mI);
        Ecur = Enext;
    };
    if constexpr (PAIR) {
#pragma unroll 1
        for (int item = 0; item < nitems; item += 2) { item_body(item, IntC<1>{}); item_body(item + 1, IntC<2>{}); }
    } else {
#pragma unroll 1
        for (int item = 0; item < nitems; ++item) item_body(item, IntC<0>{});
    }
    if constexpr (STATS != 0) {
        stats_fold();
        __syncthreads();                                                 // (the tile buffers are free)
        double* sred = reinterpret_cast<double*>(ldsb + L::TILE_OFF);    // [8 waves][2][16]
        if (i == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { sred[(wave * 2 + 0) * 16 + 4 * g + j] = da1[j]; sred[(wave * 2 + 1) * 16 + 4 * g + j] = da2[j]; }
        }
        __syncthreads();
        double a = 0.0, b = 0.0;
        if ((int)threadIdx.x < 16) {
#pragma unroll
            for (int w = 0; w < 8; ++w) { a += sred[(w * 2 + 0) * 16 + threadIdx.x]; b += sred[(w * 2 + 1) * 16 + threadIdx.x]; }
        }
        stats_rows(a, b);
    }
}

template <int STATS, bool PRO, bool PAIR>
int launch_fwdsp8(const FwdP& p, int gy, hipStream_t st) {
    constexpr int shm = Sp8Lds::TOTAL_B;
    auto kern = conv3_fwdsp8_kernel<STATS, PRO, PAIR>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, shm);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.nblocks / 2, gy), dim3(512), shm, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}

template <int NREP, int STATS, bool PRO, bool PAIR, bool WG3 = false>
int launch_fwdsp(const FwdP& p, int gy, hipStream_t st) {
    constexpr int shm = SpLds<NREP, STATS, WG3>::TOTAL_B;
    auto kern = conv3_fwdsp_kernel<NREP, STATS, PRO, PAIR, WG3>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, shm);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    { static int occ = -1; if (occ < 0) { occ = getenv("DA_OCC") ? 1 : 0; if (occ) { int nb = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, shm); fprintf(stderr, "[occ] fwdsp<%d,S%d,PRO%d,PAIR%d> lds %d B -> %d workgroups per CU\n", NREP, STATS, (int)PRO, (int)PAIR, shm, nb); } } }
    hipLaunchKernelGGL(kern, dim3(p.nblocks, gy), dim3(256), shm, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}

}  // namespace

bool da_conv3_fwdsp_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("DA_FWDSP"); on = (e && atoi(e)) ? 1 : 0; }      // off by default: measured equal alone and 0.1 - 0.3 ms SLOWER in the seg step
    return on != 0;
}

int da_conv3_fwdsp_launch(const DaC3FwdP& p, int gy, int nrep, int stats, int pro, int pair, hipStream_t st) {
    static int w8 = -1; if (w8 < 0) { const char* e = getenv("DA_FWDSP8"); w8 = (e && atoi(e)) ? 1 : 0; }
    if (w8 && nrep == 1 && p.nblocks >= 16 && p.nblocks % 16 == 0) {      // eight-wave pipelined form: half as many, twice as large workgroups (multiple of 8: tile_walk's XCD split)
#define DA_FSP8(s, pr, pa) if (stats == s && (pro != 0) == pr && (pair != 0) == pa) return launch_fwdsp8<s, pr, pa>(p, gy, st)
        DA_FSP8(0, false, false); DA_FSP8(0, false, true); DA_FSP8(0, true, false);
        DA_FSP8(1, false, false); DA_FSP8(1, false, true); DA_FSP8(1, true, false);
        DA_FSP8(2, false, false);
#undef DA_FSP8
    }
    { static int wg3 = -1; if (wg3 < 0) { const char* e = getenv("DA_FWD_WG3"); wg3 = (e && atoi(e)) ? 1 : 0; }
      if (wg3 && nrep == 1 && stats != 2) {      // experiment: three workgroups per CU (<= 168 registers, one weight buffer, no paired staging); the caller sized the grid
#define DA_FSP3(s, pr) if (stats == s && (pro != 0) == pr) return launch_fwdsp<1, s, pr, false, true>(p, gy, st)
          DA_FSP3(0, false); DA_FSP3(0, true); DA_FSP3(1, false); DA_FSP3(1, true);
#undef DA_FSP3
      } }
#define DA_FSP(nr, s, pr, pa) if (nrep == nr && stats == s && (pro != 0) == pr && (pair != 0) == pa) return launch_fwdsp<nr, s, pr, pa>(p, gy, st)
    DA_FSP(1, 0, false, false); DA_FSP(1, 0, false, true); DA_FSP(1, 0, true, false);
    DA_FSP(1, 1, false, false); DA_FSP(1, 1, false, true); DA_FSP(1, 1, true, false);
    DA_FSP(1, 2, false, false);
    DA_FSP(2, 0, false, false); DA_FSP(2, 0, true, false);
    DA_FSP(2, 1, false, false); DA_FSP(2, 1, true, false);
#undef DA_FSP
    return DA_ERR_UNSUPPORTED;
}
