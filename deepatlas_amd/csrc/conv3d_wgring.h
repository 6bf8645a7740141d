// Split-mode weight gradient, fourth form: the eight-wave 16-channel kernel of conv3d_mfma.hip (conv3_split_wgrad16_kernel) walking z COLUMNS
// with the x halo planes in an LDS RING.  Included by conv3d_mfma.hip inside its anonymous namespace (WgP, the staging helpers, split_f16.h).
//
// Why.  conv3_split_wgrad16_kernel stages a 4 x 10 x 18 halo box for every 2 x 8 x 16 output tile: 2.81 x the tile's voxels, every tile, through the
// L1 -- 7.1 GB per launch of the 48 -> 16 layer against 2.5 GB of tensors, at the ~10 B / clk / CU an L1 sustains on 64-byte sector requests
// (profiles/r04_wgrad_memory_path.txt: the staging loads ALONE take the row-owner kernel's whole time).  A weight gradient keeps its sums per
// (tap, cin, cout) and only walks voxels, so a workgroup can walk a z column: consecutive tiles (z0, z0 + 2, ...) share two of their four halo
// planes.  Here the x planes live in a ring of six plane slots (three slabs of two planes): a tile reads the slabs (lower, upper), the next
// tile's new slab (planes z0 + 3, z0 + 4) is converted into the third slot while the current tile's slabs are still in use -- one barrier per tile
// instead of two, 3 + 2 parked quads per thread instead of 6 + 2 -- and only 1.41 x the tile's voxels are loaded per tile: 4.5 GB per launch.
// A run starts with a PRIME: the tile's lower slab alone (its maximum fixes the ring's scale), then the upper slab + dY as an ordinary unit of the
// pipeline with nothing to multiply beside it.  Register budget: <= 200, so that two 56-register waves of the HBM-bound BatchNorm-backward kernels
// fit per SIMD beside the two resident waves of this kernel (in the training step this kernel runs on the side stream beside them; a form with
// specialised producer / consumer waves -- 1.51 instead of 1.61 ms alone on the 48 -> 16 layer, 241 registers -- lost 0.3 ms of step time to exactly
// that and was removed: git history of this file).
//
// Scales (split_f16.h).  The planes of a ring are shared by consecutive tiles, so they carry ONE power-of-two scale 2^ea for the whole run of
// tiles; a run ends at the end of a column, at the end of the workgroup's range, or when the incoming slab does not fit the scale: it would
// overflow fp16 (its ideal exponent e_new < ea) or the tile's own window {upper slab, new slab} would sit more than 2^3 below the ideal scale
// (ea < min(e_up, e_new) - 3).  The next run starts with a whole-tile load and a fresh scale.  What a tile's x operands get is therefore the
// same as in the row-owner / eight-wave kernels, whose accumulator-unit hysteresis also stages x up to 2^3 below its box's ideal scale.  dY has
// no halo: its scale is per tile, 2^(E - ea) with E the accumulators' unit (kept while the ideal unit lies within [E, E + 3]).
template <bool PRO>
__global__ void __launch_bounds__(512, 1) conv3_split_wgrad16r_kernel(WgP p) {
    using WFrag = f16x8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CK = 16, CG = 16, TVOX = 2 * TY * TX, NT = 512;
    constexpr int ZPQ = DA_WG16_ZPAD, ZPE = 4 * ZPQ;
    constexpr int PV = HY * HX;                               // voxels of a halo plane
    constexpr int PS = PV * 8 + ZPE;                          // two-byte elements of one plane slot of one half image
    constexpr int NSLOT = 6;                                  // plane slots: three slabs of two planes
    constexpr int PLH = NSLOT * PS, PLA = 2 * PLH;            // elements per half image / per fp16 plane (h | l)
    constexpr int PLY = TVOX * CG;                            // elements per fp16 plane of a dY tile (two buffers of two planes)
    constexpr int QA = CK / 4, QY = CG / 4;
    constexpr int NIT2 = (2 * PV * QA + NT - 1) / NT;         // parked quads of a slab (3)
    constexpr int NITY = (TVOX * QY + NT - 1) / NT;           // ... of a dY tile (2)
    float* ldsA = lds;
    float* ldsY = lds + PLA;                                  // (two planes of PLA two-byte elements = PLA floats)
    float* smax = ldsY + 2 * PLY;                             // [2 parities][3: x lower | x upper (or the slab) | dY][8 waves]
    typedef s16x4 __attribute__((address_space(3))) * lds_frag_ptr;
    const short* ldsAh = reinterpret_cast<const short*>(ldsA);
    const short* ldsYh = reinterpret_cast<const short*>(ldsY);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wave & 3, wh = wave >> 2;                  // row pair, channel half
    const int i = lane & 15, g = lane >> 4, q = i & 3, vq = i >> 2;
    const int slab = blockIdx.x, nsl = gridDim.x, ch = blockIdx.y, cg = blockIdx.z;
    const int cbase = ch * CK;
    const float* src; int Cs, choff;
    if (cbase < p.C1) { src = p.in1; Cs = p.C1; choff = cbase; } else { src = p.in2; Cs = p.C2; choff = cbase - p.C1; }
    const int c4 = (int)threadIdx.x % QA;
    unsigned vmA = 0;
    // PRO: the per-channel scale / shift quads of the input prologue live in LDS (eight registers less: the budget above), re-read where they are applied
    float* spro = smax + 2 * 24;                                // [scale | shift][4 channel quads]
    float pslope = -1.f;
    if constexpr (PRO) {
        if (threadIdx.x < 8) {
            const int cofs = choff + ((int)threadIdx.x & 3) * 4;
            const float* sp = threadIdx.x < 4 ? (cbase < p.C1 ? p.ps1 : p.ps2) : (cbase < p.C1 ? p.pt1 : p.pt2);
            reinterpret_cast<float4*>(spro)[threadIdx.x] = *reinterpret_cast<const float4*>(sp + cofs);
        }
        pslope = cbase < p.C1 ? p.pslope1 : p.pslope2;
        __syncthreads();
    }
    // fragment sources: the eight-wave kernel's, with the z plane of a read resolved through the ring (offC, per tile)
    const int laneA0 = ((2 * wr) * HX + 8 * (g & 1) + vq) * 8 + (q & 1) * 4 + wh * PLH;
    const int laneY = ((((g >> 1) * TY) + 2 * wr) * TX + 8 * (g & 1) + vq) * CG + q * 4;
    int offC[5];
    auto set_ring = [&](int L, int U) {                        // a tile's planes 0, 1 = slab slot L, planes 2, 3 = slab slot U
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int combo = (c < 4) ? 2 * c + (q >> 1) : 8;
            const int rel = (g >> 1) + combo / 3;              // plane of the tile's halo box this lane reads for tap column `combo`
            const int ps = rel < 2 ? 2 * L + rel : 2 * U + rel - 2;
            offC[c] = ps * PS + (combo % 3) * 8;
        }
    };
    auto tr8 = [&](const short* a, int step) -> WFrag {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)a);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)(a + step));
        return __builtin_bit_cast(WFrag, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma = [&](f32x4 c, const WFrag& a, const WFrag& b) -> f32x4 { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); };
    struct F3 { WFrag p[2]; };
    auto loadF = [&](int c, int h) -> F3 {
        F3 f; const short* a = ldsAh + laneA0 + offC[c] + h * (HX * 8);
        f.p[0] = tr8(a, 4 * 8); f.p[1] = tr8(a + PLA, 4 * 8);
        return f;
    };
    auto loadG = [&](int h) -> F3 {
        F3 f; const short* a = ldsAh + laneA0 + offC[4] + (h + (q >> 1)) * (HX * 8);
        f.p[0] = tr8(a, 4 * 8); f.p[1] = tr8(a + PLA, 4 * 8);
        return f;
    };
    int ybo = 0;                                               // element offset of the dY buffer the current tile reads
    auto loadY = [&](int r) -> F3 {
        F3 f; const short* a = ldsYh + ybo + laneY + r * (TX * CG);
        f.p[0] = tr8(a, 4 * CG); f.p[1] = tr8(a + PLY, 4 * CG);
        return f;
    };
    f32x4 acc[5][3];
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 preA[NIT2], preY[NITY];
    // staging maps (launch constants): x quad idx = threadIdx.x + 512 it -> (plane, hy, hx) of a slab (two planes)
    int voA[NIT2]; unsigned pkA[NIT2];
#pragma unroll
    for (int it = 0; it < NIT2; ++it) {
        const int hv = ((int)threadIdx.x + it * NT) / QA;
        const int hx = hv % HX, t = hv / HX, hy = t % HY, hz = t / HY;
        pkA[it] = (unsigned)hz << 16 | (unsigned)hy << 8 | (unsigned)hx;
        voA[it] = (hz * p.H + hy) * p.W + hx;
    }
    auto pro_apply = [&]() {
        if constexpr (PRO) {
            const float4 psc = reinterpret_cast<const float4*>(spro)[c4], psf = reinterpret_cast<const float4*>(spro)[4 + c4];
            stage_pro_apply<0, NIT2>(preA, vmA, psc, psf, pslope);
        }
    };
    const bool smallA = (long long)4 * p.H * p.W < (1ll << 24) && (long long)Cs * 4 < (1ll << 24);
    const int yq4 = cg * CG + ((int)threadIdx.x % QY) * 4;
    const int yv0 = (int)threadIdx.x / QY;
    int voY[NITY];
    const bool smallY = (long long)2 * p.H * p.W < (1ll << 24) && (long long)p.Cout * 4 < (1ll << 24);
#pragma unroll
    for (int it = 0; it < NITY; ++it) { const int v = yv0 + it * (NT / QY); voY[it] = ((v >> 7) * p.H + ((v >> 4) & 7)) * p.W + (v & 15); }
    // the slab of planes zf, zf + 1 of sample n (zf may be -1, zf + 1 may reach D: zeros), halo box origin (y0 - 1, x0 - 1)
    auto issue_x = [&](int n, int zf, int y0, int x0) {
        constexpr int NPL = 2;
        const long long sample = (long long)p.D * p.H * p.W * Cs;
        const __amdgpu_buffer_rsrc_t rs = da_rsrc_n<false>(src, n, sample);
        const bool interior = smallA && zf >= 0 && zf + NPL <= p.D && y0 >= 1 && y0 + HY - 2 < p.H && x0 >= 1 && x0 + HX - 2 < p.W;
        const unsigned Cs4 = (unsigned)Cs * 4u, cofs4 = (unsigned)(choff + c4 * 4) * 4u;
        const unsigned base = (unsigned)((zf * p.H + (y0 - 1)) * p.W + (x0 - 1)) * Cs4 + cofs4;
        if constexpr (PRO) vmA = 0;
#pragma unroll
        for (int it = 0; it < NIT2; ++it) {
            const int hz = (int)(pkA[it] >> 16), hy = (int)((pkA[it] >> 8) & 255u), hx = (int)(pkA[it] & 255u);
            unsigned so;
            if (interior) so = ((it + 1) * NT <= NPL * PV * QA || hz < NPL) ? __umul24((unsigned)voA[it], Cs4) + base : 0xFFFFFFFFu;
            else {
                const int z = zf + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
                const bool inb = hz < NPL && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                so = inb ? (unsigned)((z * p.H + y) * p.W + x) * Cs4 + cofs4 : 0xFFFFFFFFu;
            }
            preA[it] = da_buf_load4(rs, so);
            if constexpr (PRO) vmA |= (so != 0xFFFFFFFFu ? 1u : 0u) << it;
        }
    };
    auto issue_y = [&](int n, int z0, int y0, int x0) {
        const long long sampleY = (long long)p.D * p.H * p.W * p.Cout;
        const __amdgpu_buffer_rsrc_t ry = da_rsrc_n<false>(p.dy, n, sampleY);
        const bool inside = smallY && z0 + 2 <= p.D && y0 + TY <= p.H && x0 + TX <= p.W && cg * CG + CG <= p.Cout;
        const unsigned baseY = ((unsigned)((z0 * p.H + y0) * p.W + x0) * (unsigned)p.Cout + (unsigned)yq4) * 4u;
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
            const int v = yv0 + it * (NT / QY);
            const int vx = v & 15, vy = (v >> 4) & 7, vz = v >> 7;
            unsigned off;
            if (inside) off = __umul24((unsigned)voY[it], (unsigned)p.Cout * 4u) + baseY;
            else {
                const int x = x0 + vx, y = y0 + vy, z = z0 + vz;
                const bool vin = z < p.D && y < p.H && x < p.W && yq4 < p.Cout;
                off = vin ? ((unsigned)((z * p.H + y) * p.W + x) * (unsigned)p.Cout + (unsigned)yq4) * 4u : 0xFFFFFFFFu;
            }
            preY[it] = da_buf_load4(ry, off);
        }
    };
    // parked quads -> the two fp16 planes of the ring, the slab's planes going to plane slots ps0, ps0 + 1
    auto write_x = [&](int ps0, float sa) {
#pragma unroll
        for (int it = 0; it < NIT2; ++it) {
            const int hz = (int)(pkA[it] >> 16);
            if ((it + 1) * NT <= 2 * PV * QA || hz < 2) {
                const int hv = ((int)threadIdx.x + it * NT) >> 2;
                const int idx = (c4 >> 1) * (PLH / 4) + (ps0 + hz) * (PS / 4) + (hv - hz * PV) * 2 + (c4 & 1);
                uint2 h, l; da_split2(preA[it], sa, h, l);
                reinterpret_cast<uint2*>(ldsA)[idx] = h; reinterpret_cast<uint2*>(ldsA)[idx + PLA / 4] = l;
            }
        }
    };
    auto write_y = [&](int yb, float sy) {
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
            const int idx = yb * (2 * PLY / 4) + (int)threadIdx.x + it * NT;
            uint2 h, l; da_split2(preY[it], sy, h, l);
            reinterpret_cast<uint2*>(ldsY)[idx] = h; reinterpret_cast<uint2*>(ldsY)[idx + PLY / 4] = l;
        }
    };
    auto wave_read_max = [&](const float* s) -> float {       // the eight waves' maxima of one quantity -> a wave-uniform float
        const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
        const float m = fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
        return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m)));
    };
    // the unit 2^-E of a tile's products given the ring's x exponent and the tile's ideal dY exponent (hysteresis and cap as in the other forms)
    int Eacc = 0, Emin = 0; bool first_tile = true;
    auto pick_E = [&](int ea, int ey_ideal, int Eref) -> int {
        int E = ea + ey_ideal;
        if (!first_tile) E = min(E, Emin + 40);
        if (!first_tile && E >= Eref && E <= Eref + 3) E = Eref;
        Emin = first_tile ? E : min(Emin, E);
        first_tile = false;
        return E;
    };

    const int lo = slab * p.tiles_per_slab, hi = min(lo + p.tiles_per_slab, p.ntiles);
    int pos = lo;
    int par = 0;
    int hint = kSplitEmax;                                     // upper bound for the next run's ring exponent: the ideal exponent of the slab that ended the previous run
#pragma unroll 1
    while (pos < hi) {
        // ---- start of a run at tile `pos`.  Prime: its lower slab (planes z0 - 1, z0) alone; the ring's scale is that slab's ideal one, capped by `hint`
        // position -> tile, z fastest (a workgroup's contiguous range walks whole columns): three scalar divisions per RUN, none per tile, no tile table
        const int tz0 = pos % p.ntz, col = pos / p.ntz, tx0 = col % p.ntx, rr = col / p.ntx;
        const int n = rr / p.nty, zc = 2 * tz0, y0 = (rr % p.nty) * TY, x0 = tx0 * TX;
        issue_x(n, zc - 1, y0, x0);
        pro_apply();
        {
            const float a0 = da_wave_max_nonneg(stage_absmax<NIT2>(preA));
            if (lane == 0) smax[par * 24 + wave] = a0;
        }
        __syncthreads();
        int e_up = da_scale_exp(wave_read_max(smax + par * 24));
        const int ea = min(e_up, hint);
        hint = kSplitEmax;
        write_x(0, da_pow2(ea));
        par ^= 1;
        // its upper slab (planes z0 + 1, z0 + 2) and dY tile: an ordinary unit of the pipeline, with no tile to multiply beside it yet
        issue_x(n, zc + 1, y0, x0);
        issue_y(n, zc, y0, x0);
        pro_apply();
        {
            const float a1 = da_wave_max_nonneg(stage_absmax<NIT2>(preA)), my = da_wave_max_nonneg(stage_absmax<NITY>(preY));
            if (lane == 0) { smax[par * 24 + 8 + wave] = a1; smax[par * 24 + 16 + wave] = my; }
        }
        __syncthreads();
        int Ecur;
        {
            const int e_new = da_scale_exp(wave_read_max(smax + par * 24 + 8));
            if (!(ea <= e_new && ea >= min(e_up, e_new) - 3)) { hint = e_new; par ^= 1; continue; }      // (the retry's scale min(e_lo, e_new) fits both slabs)
            Ecur = pick_E(ea, da_scale_exp(wave_read_max(smax + par * 24 + 16)), Eacc);
            write_x(2, da_pow2(ea));
            write_y(0, da_pow2(Ecur - ea));
            e_up = e_new;
        }
        par ^= 1;
        int L = 0, U = 1, yb = 0, t = pos;
        // the next tile of the column, if it is ours: its new slab (planes z + 3, z + 4 of the current tile) and its dY
        bool have_next = pos + 1 < hi && tz0 + 1 < p.ntz;
        if (have_next) {
            issue_x(n, zc + 3, y0, x0);
            issue_y(n, zc + 2, y0, x0);
            pro_apply();
            const float a1 = da_wave_max_nonneg(stage_absmax<NIT2>(preA)), my = da_wave_max_nonneg(stage_absmax<NITY>(preY));
            if (lane == 0) { smax[par * 24 + 8 + wave] = a1; smax[par * 24 + 16 + wave] = my; }
        }
        __syncthreads();
        // ---- the run: one barrier per tile
#pragma unroll 1
        for (;;) {
            int Enext = Ecur; bool conv = false, have_next2 = false;
            if (have_next) {
                const int e_new = da_scale_exp(wave_read_max(smax + par * 24 + 8));
                const int ey = da_scale_exp(wave_read_max(smax + par * 24 + 16));
                conv = ea <= e_new && ea >= min(e_up, e_new) - 3;
                if (!conv) hint = e_new;                                        // (the next run starts at tile t + 1: its lower slab is the current upper one)
                if (conv) {
                    const int N = U == 2 ? 0 : U + 1;
                    Enext = pick_E(ea, ey, Ecur);
                    write_x(2 * N, da_pow2(ea));
                    write_y(yb ^ 1, da_pow2(Enext - ea));
                    e_up = e_new;
                    have_next2 = t + 2 < hi && tz0 + (t + 2 - pos) < p.ntz;
                    if (have_next2 && !(p.ablate & 1)) {
                        const int z2 = zc + 2 * (t - pos);                       // z0 of tile t
                        issue_x(n, z2 + 5, y0, x0);
                        issue_y(n, z2 + 4, y0, x0);
                    }
                }
            }
            if (Ecur != Eacc) {
                const float f = da_acc_factor(Ecur - Eacc);
#pragma unroll
                for (int c = 0; c < 5; ++c)
#pragma unroll
                    for (int d = 0; d < 3; ++d) acc[c][d] = acc[c][d] * f;
                Eacc = Ecur;
            }
            set_ring(L, U);
            ybo = yb * (2 * PLY);
            if (!(p.ablate & 2)) {
                F3 Y0 = loadY(0), Y1 = loadY(1);
                F3 Fa = loadF(0, 0), Fb = loadF(0, 1), Fc = loadF(0, 2), Fd, Na, Nb, Nc;
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};             // (x, dY) plane pairs, small terms first
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    Fd = loadF(c, 3);
                    Na = (c < 3) ? loadF(c + 1, 0) : loadG(0);
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr) {
                        acc[c][0] = mma(acc[c][0], Fa.p[PA[pr]], Y0.p[PB[pr]]);
                        acc[c][1] = mma(acc[c][1], Fb.p[PA[pr]], Y0.p[PB[pr]]);
                        acc[c][2] = mma(acc[c][2], Fc.p[PA[pr]], Y0.p[PB[pr]]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (c < 3) { Nb = loadF(c + 1, 1); Nc = loadF(c + 1, 2); } else { Nb = loadG(1); Nc = loadF(4, 2); }
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr) {
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
                    for (int pr = 0; pr < 3; ++pr) {
                        acc[4][0] = mma(acc[4][0], Fa.p[PA[pr]], Y0.p[PB[pr]]);
                        acc[4][1] = mma(acc[4][1], Fc.p[PA[pr]], Y0.p[PB[pr]]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr) {
                        acc[4][0] = mma(acc[4][0], Fb.p[PA[pr]], Y1.p[PB[pr]]);
                        acc[4][1] = mma(acc[4][1], Fd.p[PA[pr]], Y1.p[PB[pr]]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (conv && have_next2) {                                           // the slab after next has landed: its maxima, for the decision at the top of the next tile
                pro_apply();
                const float a1 = da_wave_max_nonneg(stage_absmax<NIT2>(preA)), my = da_wave_max_nonneg(stage_absmax<NITY>(preY));
                if (lane == 0) { smax[(par ^ 1) * 24 + 8 + wave] = a1; smax[(par ^ 1) * 24 + 16 + wave] = my; }
            }
            __syncthreads();
            if (!conv) break;
            L = U; U = (U == 2 ? 0 : U + 1); yb ^= 1; t += 1; have_next = have_next2; par ^= 1; Ecur = Enext;
        }
        pos = t + 1;
    }
    {
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
    (void)nsl;
}

// ---------------------------------------------------------------------------------------------------
// The same walk for bf16 ACTIVATION STORAGE in the bf16 matrix mode (BASELINE configs[4]): x and dY are bf16 tensors, one bf16 plane in LDS, one
// v_mfma_f32_16x16x32_bf16 per product -- no scales, so a run only ends with its column or the workgroup's range, and the ring's bookkeeping reduces to
// the slot rotation.  Without the input prologue the eight bytes of a loaded quad go untouched into the LDS image.  A third of the matrix work of the
// split mode on the same bytes / 2: this form is bound by its staging, which is what the ring halves.
// ---------------------------------------------------------------------------------------------------
template <bool PRO>
__global__ void __launch_bounds__(512, 1) conv3_bf16_wgrad16r_kernel(WgP p) {
    using WFrag = bf16x8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CK = 16, CG = 16, TVOX = 2 * TY * TX, NT = 512;
    constexpr int ZPQ = DA_WG16_ZPAD, ZPE = 4 * ZPQ;
    constexpr int PV = HY * HX, PS = PV * 8 + ZPE, NSLOT = 6, PLH = NSLOT * PS, PLA = 2 * PLH, PLY = TVOX * CG;
    constexpr int QA = CK / 4, QY = CG / 4;
    constexpr int NIT2 = (2 * PV * QA + NT - 1) / NT, NITY = (TVOX * QY + NT - 1) / NT;
    constexpr bool RAWA = !PRO;
    float* ldsA = lds;                                          // one plane of PLA two-byte elements = PLA / 2 floats
    float* ldsY = lds + PLA / 2;                                // two buffers of PLY elements
    float* spro = ldsY + PLY;                                   // [scale | shift][4 channel quads]
    typedef s16x4 __attribute__((address_space(3))) * lds_frag_ptr;
    const short* ldsAh = reinterpret_cast<const short*>(ldsA);
    const short* ldsYh = reinterpret_cast<const short*>(ldsY);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wave & 3, wh = wave >> 2;
    const int i = lane & 15, g = lane >> 4, q = i & 3, vq = i >> 2;
    const int slab = blockIdx.x, ch = blockIdx.y, cg = blockIdx.z;
    const int cbase = ch * CK;
    const float* src; int Cs, choff;
    if (cbase < p.C1) { src = p.in1; Cs = p.C1; choff = cbase; } else { src = p.in2; Cs = p.C2; choff = cbase - p.C1; }
    const int c4 = (int)threadIdx.x % QA;
    unsigned vmA = 0;
    float pslope = -1.f;
    if constexpr (PRO) {
        if (threadIdx.x < 8) {
            const int cofs = choff + ((int)threadIdx.x & 3) * 4;
            const float* sp = threadIdx.x < 4 ? (cbase < p.C1 ? p.ps1 : p.ps2) : (cbase < p.C1 ? p.pt1 : p.pt2);
            reinterpret_cast<float4*>(spro)[threadIdx.x] = *reinterpret_cast<const float4*>(sp + cofs);
        }
        pslope = cbase < p.C1 ? p.pslope1 : p.pslope2;
        __syncthreads();
    }
    const int laneA0 = ((2 * wr) * HX + 8 * (g & 1) + vq) * 8 + (q & 1) * 4 + wh * PLH;
    const int laneY = ((((g >> 1) * TY) + 2 * wr) * TX + 8 * (g & 1) + vq) * CG + q * 4;
    int offC[5];
    auto set_ring = [&](int L, int U) {
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int combo = (c < 4) ? 2 * c + (q >> 1) : 8;
            const int rel = (g >> 1) + combo / 3;
            const int ps = rel < 2 ? 2 * L + rel : 2 * U + rel - 2;
            offC[c] = ps * PS + (combo % 3) * 8;
        }
    };
    auto tr8 = [&](const short* a, int step) -> WFrag {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)a);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_frag_ptr)(a + step));
        return __builtin_bit_cast(WFrag, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma = [&](f32x4 c, const WFrag& a, const WFrag& b) -> f32x4 { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); };
    auto loadF = [&](int c, int h) -> WFrag { return tr8(ldsAh + laneA0 + offC[c] + h * (HX * 8), 4 * 8); };
    auto loadG = [&](int h) -> WFrag { return tr8(ldsAh + laneA0 + offC[4] + (h + (q >> 1)) * (HX * 8), 4 * 8); };
    int ybo = 0;
    auto loadY = [&](int r) -> WFrag { return tr8(ldsYh + ybo + laneY + r * (TX * CG), 4 * CG); };
    f32x4 acc[5][3];
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 preA[NIT2], preY[NITY];
    int voA[NIT2]; unsigned pkA[NIT2];
#pragma unroll
    for (int it = 0; it < NIT2; ++it) {
        const int hv = ((int)threadIdx.x + it * NT) / QA;
        const int hx = hv % HX, t = hv / HX, hy = t % HY, hz = t / HY;
        pkA[it] = (unsigned)hz << 16 | (unsigned)hy << 8 | (unsigned)hx;
        voA[it] = (hz * p.H + hy) * p.W + hx;
    }
    const bool smallA = (long long)4 * p.H * p.W < (1ll << 24) && (long long)Cs * 2 < (1ll << 24);
    const int yq4 = cg * CG + ((int)threadIdx.x % QY) * 4;
    const int yv0 = (int)threadIdx.x / QY;
    int voY[NITY];
    const bool smallY = (long long)2 * p.H * p.W < (1ll << 24) && (long long)p.Cout * 2 < (1ll << 24);
#pragma unroll
    for (int it = 0; it < NITY; ++it) { const int v = yv0 + it * (NT / QY); voY[it] = ((v >> 7) * p.H + ((v >> 4) & 7)) * p.W + (v & 15); }
    auto issue_x = [&](int n, int zf, int y0, int x0) {
        const __amdgpu_buffer_rsrc_t rs = da_rsrc_n<true>(src, n, (long long)p.D * p.H * p.W * Cs);
        const bool interior = smallA && zf >= 0 && zf + 2 <= p.D && y0 >= 1 && y0 + HY - 2 < p.H && x0 >= 1 && x0 + HX - 2 < p.W;
        const unsigned Cs4 = (unsigned)Cs * 2u, cofs4 = (unsigned)(choff + c4 * 4) * 2u;
        const unsigned base = (unsigned)((zf * p.H + (y0 - 1)) * p.W + (x0 - 1)) * Cs4 + cofs4;
        if constexpr (PRO) vmA = 0;
#pragma unroll
        for (int it = 0; it < NIT2; ++it) {
            const int hz = (int)(pkA[it] >> 16), hy = (int)((pkA[it] >> 8) & 255u), hx = (int)(pkA[it] & 255u);
            unsigned so;
            if (interior) so = ((it + 1) * NT <= 2 * PV * QA || hz < 2) ? __umul24((unsigned)voA[it], Cs4) + base : 0xFFFFFFFFu;
            else {
                const int z = zf + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
                const bool inb = hz < 2 && (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                so = inb ? (unsigned)((z * p.H + y) * p.W + x) * Cs4 + cofs4 : 0xFFFFFFFFu;
            }
            preA[it] = da_buf_loadq<true, RAWA>(rs, so);
            if constexpr (PRO) vmA |= (so != 0xFFFFFFFFu ? 1u : 0u) << it;
        }
    };
    auto issue_y = [&](int n, int z0, int y0, int x0) {
        const __amdgpu_buffer_rsrc_t ry = da_rsrc_n<true>(p.dy, n, (long long)p.D * p.H * p.W * p.Cout);
        const bool inside = smallY && z0 + 2 <= p.D && y0 + TY <= p.H && x0 + TX <= p.W && cg * CG + CG <= p.Cout;
        const unsigned baseY = ((unsigned)((z0 * p.H + y0) * p.W + x0) * (unsigned)p.Cout + (unsigned)yq4) * 2u;
#pragma unroll
        for (int it = 0; it < NITY; ++it) {
            const int v = yv0 + it * (NT / QY);
            const int vx = v & 15, vy = (v >> 4) & 7, vz = v >> 7;
            unsigned off;
            if (inside) off = __umul24((unsigned)voY[it], (unsigned)p.Cout * 2u) + baseY;
            else {
                const int x = x0 + vx, y = y0 + vy, z = z0 + vz;
                const bool vin = z < p.D && y < p.H && x < p.W && yq4 < p.Cout;
                off = vin ? ((unsigned)((z * p.H + y) * p.W + x) * (unsigned)p.Cout + (unsigned)yq4) * 2u : 0xFFFFFFFFu;
            }
            preY[it] = da_buf_loadq<true, true>(ry, off);
        }
    };
    auto pack = [&](const float4 v, bool raw) -> uint2 {
        return raw ? make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)) : make_uint2(da_bf16x2(v.x, v.y), da_bf16x2(v.z, v.w));
    };
    auto write_x = [&](int ps0) {
        if constexpr (PRO) {
            const float4 psc = reinterpret_cast<const float4*>(spro)[c4], psf = reinterpret_cast<const float4*>(spro)[4 + c4];
            stage_pro_apply<0, NIT2>(preA, vmA, psc, psf, pslope);
        }
#pragma unroll
        for (int it = 0; it < NIT2; ++it) {
            const int hz = (int)(pkA[it] >> 16);
            if ((it + 1) * NT <= 2 * PV * QA || hz < 2) {
                const int hv = ((int)threadIdx.x + it * NT) >> 2;
                reinterpret_cast<uint2*>(ldsA)[(c4 >> 1) * (PLH / 4) + (ps0 + hz) * (PS / 4) + (hv - hz * PV) * 2 + (c4 & 1)] = pack(preA[it], RAWA);
            }
        }
    };
    auto write_y = [&](int yb) {
#pragma unroll
        for (int it = 0; it < NITY; ++it) reinterpret_cast<uint2*>(ldsY)[yb * (PLY / 4) + (int)threadIdx.x + it * NT] = pack(preY[it], true);
    };
    const int lo = slab * p.tiles_per_slab, hi = min(lo + p.tiles_per_slab, p.ntiles);
    int pos = lo;
#pragma unroll 1
    while (pos < hi) {
        const int tz0 = pos % p.ntz, col = pos / p.ntz, tx0 = col % p.ntx, rr = col / p.ntx;
        const int n = rr / p.nty, zc = 2 * tz0, y0 = (rr % p.nty) * TY, x0 = tx0 * TX;
        __syncthreads();                                         // every wave is done with the previous run's ring
        issue_x(n, zc - 1, y0, x0);
        write_x(0);
        issue_x(n, zc + 1, y0, x0);
        issue_y(n, zc, y0, x0);
        write_x(2);
        write_y(0);
        int L = 0, U = 1, yb = 0, t = pos;
        bool have_next = pos + 1 < hi && tz0 + 1 < p.ntz;
        if (have_next) { issue_x(n, zc + 3, y0, x0); issue_y(n, zc + 2, y0, x0); }
        __syncthreads();
#pragma unroll 1
        for (;;) {
            bool have_next2 = false;
            const int N = U == 2 ? 0 : U + 1;
            if (have_next) {
                write_x(2 * N);
                write_y(yb ^ 1);
                have_next2 = t + 2 < hi && tz0 + (t + 2 - pos) < p.ntz;
                if (have_next2 && !(p.ablate & 1)) {
                    const int z2 = zc + 2 * (t - pos);
                    issue_x(n, z2 + 5, y0, x0);
                    issue_y(n, z2 + 4, y0, x0);
                }
            }
            set_ring(L, U);
            ybo = yb * PLY;
            if (!(p.ablate & 2)) {
                const WFrag Y0 = loadY(0), Y1 = loadY(1);
                WFrag Fa = loadF(0, 0), Fb = loadF(0, 1), Fc = loadF(0, 2), Fd, Na, Nb, Nc;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    Fd = loadF(c, 3);
                    Na = (c < 3) ? loadF(c + 1, 0) : loadG(0);
                    acc[c][0] = mma(acc[c][0], Fa, Y0); acc[c][1] = mma(acc[c][1], Fb, Y0); acc[c][2] = mma(acc[c][2], Fc, Y0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (c < 3) { Nb = loadF(c + 1, 1); Nc = loadF(c + 1, 2); } else { Nb = loadG(1); Nc = loadF(4, 2); }
                    acc[c][0] = mma(acc[c][0], Fb, Y1); acc[c][1] = mma(acc[c][1], Fc, Y1); acc[c][2] = mma(acc[c][2], Fd, Y1);
                    __builtin_amdgcn_sched_barrier(0);
                    Fa = Na; Fb = Nb; Fc = Nc;
                }
                Fd = loadF(4, 3);
                acc[4][0] = mma(acc[4][0], Fa, Y0); acc[4][1] = mma(acc[4][1], Fc, Y0);
                __builtin_amdgcn_sched_barrier(0);
                acc[4][0] = mma(acc[4][0], Fb, Y1); acc[4][1] = mma(acc[4][1], Fd, Y1);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            if (!have_next) break;
            L = U; U = N; yb ^= 1; t += 1; have_next = have_next2;
        }
        pos = t + 1;
    }
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
