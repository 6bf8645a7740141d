// Fused segmentation head + softmax + Dice:  loss = Dice(softmax(x W + b), labels)  without ever writing the logits.
//
// Reference: the 1x1x1 output convolution (unets.py:249-250) followed by DiceLossMultiClass.forward with softmax=True and an index
// target (lib/loss.py:410-476).  At 160x192x160 with 32 classes the logits are the largest tensor of the step (629 MB per volume) and
// the op-by-op path moves them six times (head forward write, Dice forward read, Dice backward read + gradient write, head data
// gradient read, head weight gradient read).  The head is a 16 -> 32 channel GEMM -- cheaper to recompute than to store:
//   forward : read x (64 B / voxel), logits on the matrix cores, softmax + the three Dice sums per class in registers; nothing written
//   backward: read x again, recompute logits / softmax, form d loss / d logits from Dice's rank-structured gradient
//             g[v][c] = coef0[c] [label == c] + coef1[c], and feed it straight into the data-gradient GEMM (dx written once), the
//             weight-gradient GEMM and the bias gradient (per-workgroup partials, reduced in a fixed order).
// HBM-bound: algorithmic bytes forward (4 Cin + 1) per voxel, backward (8 Cin + 1) per voxel.
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32), operand layouts as in pointwise_mfma.hip (A rows from global in fragment order).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kHdBlocks = 1024;           // workgroups per sample (forward partial sums) / in total (backward partials)
constexpr int MT = 4;                     // M-tiles (16 voxels) per wave and chunk: a workgroup covers 256 voxels per iteration
constexpr float kLog2e = 1.44269504088896341f;

struct HdP {
    const float* x; const float* ps; const float* pt; float pslope;      // input [N][V][K] (+ optional deferred BatchNorm + activation)
    const float* wp_fwd; const float* wp_bwd; const float* bias;
    const void* labels; int label_bytes;
    long long V; int N, K, C;
    double* partial;                         // fwd: [N][gridDim.x][3][C]
    const float* coef; const float* dloss;   // bwd
    float* dx; float* wpartial;              // bwd: dx [N][V][K]; per-workgroup [K*C + C] partial (dW, dbias)
    const float* pmean; double* bst;         // bwd, K = 16 with a prologue: BatchNorm-backward sums of the PRODUCER of x (its statistics pass folded in here):
                                             // bst[workgroup][2][K] = (sum dz, sum dz (x - mean)), dz = dx act'(x scale + shift)
};

__device__ __forceinline__ float hd_act01(float z, float s) { return fmaxf(z, z * s); }
__device__ __forceinline__ long long hd_label(const void* labels, int label_bytes, long long i) {
    return label_bytes == 1 ? (long long)((const unsigned char*)labels)[i] : ((const long long*)labels)[i];
}

// packed B: wp[n][c][lane][m] = B[k = 16c + 4(lane>>4) + m][j = 16n + (lane&15)];  transposed == 0: B[k][j] = w[k*Nt + j], else w[j*Kt + k]
__global__ void hd_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int K, int N, int transposed) {
    const int KC = K / 16, NT = N / 16;
    const int total = NT * KC * 256;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int m = idx & 3, lane = (idx >> 2) & 63;
        int rest = idx >> 8;
        const int c = rest % KC; const int n = rest / KC;
        const int k = 16 * c + 4 * (lane >> 4) + m, j = 16 * n + (lane & 15);
        wp[idx] = transposed ? w[(size_t)j * K + k] : w[(size_t)k * N + j];
    }
}

// One chunk = 64 voxels per wave = MT tiles of 16.  The GEMM is run TRANSPOSED, logits^T = W^T x^T: M = classes, N = voxels, so that in
// the MFMA result layout a lane holds classes 16n + 4g + reg (8 of them for 32 classes) of ONE voxel (tile t, voxel i = lane & 15).
// The softmax of a voxel is then a reduction over this lane's registers and the three other lane groups (two DPP-free xor steps),
// the voxel's label is one load per tile, and in the backward pass the gradient with respect to the logits is already in the B-operand
// layout of the data-gradient GEMM (no transpose through LDS), whose result comes out as 4 consecutive input channels per lane
// (16-byte stores).
// Round 5: both kernels were bound by their VALU instruction count and by load latency (SQ counters: 22 % issuing, 38 % parked on s_waitcnt,
// 39 % issue-stalled; profiles/r05_head_dice_counters.txt), so (1) the next chunk's input quads and labels are loaded before the current
// chunk is worked on (hd_load / hd_labels), (2) the weights and the bias carry a factor log2(e) so that the softmax is v_exp_f32 (= 2^x) on
// the logits as they stand, 1 / sum is v_rcp_f32, and (3) everything per class -- one-hot masks, Dice sums, the softmax Jacobian -- works on
// register PAIRS (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 hd_lo(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f2 hd_hi(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
__device__ __forceinline__ f2 hd_bc(float x) { return (f2){x, x}; }

template <int KC, typename T>
__device__ __forceinline__ void hd_load(const HdP& p, const T* __restrict__ xs, long long vbase, int i, int g, float4 (&a)[MT][KC]) {
    // unconditional loads (a voxel past the end reads the last voxel's quads: finite values whose probabilities are zeroed in hd_probs): a load
    // under a branch is waited for at the end of the branch, which would undo the prefetch
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const long long vox = min(vbase + t * 16 + i, p.V - 1);
#pragma unroll
        for (int c = 0; c < KC; ++c) a[t][c] = da_ldq(xs, (vox * p.K + 16 * c + 4 * g) >> 2);
    }
}
// the labels of this lane's voxels (t, i), as loaded; hd_rel turns one into `rel` = label minus this lane group's first class 4g (a hit is
// rel == 16n + reg), -1000 past the end
__device__ __forceinline__ void hd_labels(const HdP& p, long long lbase, long long vbase, int i, int g, int (&rel)[MT]) {
    // one byte per label whatever its width (the low byte of a little-endian int64 label < 256), again without a branch around the load
    // -- and nothing computed from the loaded byte here: the first use of a loaded register is where the wait for ALL earlier loads goes
#pragma unroll
    for (int t = 0; t < MT; ++t) rel[t] = (int)((const unsigned char*)p.labels)[(lbase + min(vbase + t * 16 + i, p.V - 1)) * p.label_bytes];
}
// raw label byte -> rel (see hd_labels), at the time the chunk is worked on
__device__ __forceinline__ int hd_rel(const HdP& p, int lab, long long vox, int g) { return (vox < p.V ? lab : -1000) - 4 * g; }
// one-hot of the label over this lane's classes, as 0 / 1 floats in the accumulator layout
template <int NT>
__device__ __forceinline__ void hd_mask(int rel, f2 (&mk)[NT][2]) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        mk[n][0] = (f2){rel == 16 * n ? 1.f : 0.f, rel == 16 * n + 1 ? 1.f : 0.f};
        mk[n][1] = (f2){rel == 16 * n + 2 ? 1.f : 0.f, rel == 16 * n + 3 ? 1.f : 0.f};
    }
}

// a: the raw input quads of the chunk (hd_load) -> activated in place; acc: probabilities (0 for voxels past the end).  wf / bv carry log2(e).
template <int KC, int NT, typename T = float>
__device__ __forceinline__ void hd_probs(const HdP& p, long long vbase, int i, int g, const float4 (&psc)[KC], const float4 (&psf)[KC],
                                         const float4 (&wf)[NT][KC], const float (&bv)[NT][4], float4 (&a)[MT][KC], f32x4 (&acc)[MT][NT]) {
    if (p.ps) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const float4 sc = psc[c], sf = psf[c];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                a[t][c].x = hd_act01(a[t][c].x * sc.x + sf.x, p.pslope); a[t][c].y = hd_act01(a[t][c].y * sc.y + sf.y, p.pslope);
                a[t][c].z = hd_act01(a[t][c].z * sc.z + sf.z, p.pslope); a[t][c].w = hd_act01(a[t][c].w * sc.w + sf.w, p.pslope);
                if constexpr (DaEl<T>::bf) a[t][c] = da_unpack_bf16x4(da_pack_bf16x4(a[t][c]));      // bf16 storage: what a stored activation would hold
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = (f32x4){bv[n][0], bv[n][1], bv[n][2], bv[n][3]};
    // A = W^T fragment (lane: class 16n + i, input channel 16c + 4g + m), B = x^T fragment (lane: voxel i, input channel 16c + 4g + m);
    // the MT * NT accumulators take turns (an MFMA on the accumulator of the one before it waits for its result)
#pragma unroll
    for (int c = 0; c < KC; ++c) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][c].x, a[t][c].x, acc[t][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][c].y, a[t][c].y, acc[t][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][c].z, a[t][c].z, acc[t][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][c].w, a[t][c].w, acc[t][n], 0, 0, 0);
    }
    // softmax over the classes of voxel (t, i): NT * 4 values in this lane x the four lane groups (F.softmax(source, dim=1), loss.py:426-427);
    // the logits are in units of log 2: 2^(l - max) = e^(logit - max logit)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const bool valid = vbase + t * 16 + i < p.V;
        float m = acc[t][0][0];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) m = fmaxf(m, acc[t][n][reg]);
        m = da_rows_max(m);
        f2 s2 = (f2){0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc[t][n][reg] = __builtin_amdgcn_exp2f(acc[t][n][reg] - m);      // arguments <= 0, relative error ~1e-6
            s2 += hd_lo(acc[t][n]) + hd_hi(acc[t][n]);
        }
        const float s = da_rows_sum(s2.x + s2.y);
        const f2 inv = hd_bc(valid ? __builtin_amdgcn_rcpf(s) : 0.f);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const f2 lo = hd_lo(acc[t][n]) * inv, hi = hd_hi(acc[t][n]) * inv;
            acc[t][n] = (f32x4){lo.x, lo.y, hi.x, hi.y};
        }
    }
}

// weights as the A operand: wf[n][c] (lane (i, g)) = W[input channel 16c + 4g + m][class 16n + i], m = .x .. .w  = packed B of the plain form
// bias of this lane's classes 16n + 4g + reg
template <int NT>
__device__ __forceinline__ void hd_bias(const HdP& p, int g, float (&bv)[NT][4]) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) bv[n][reg] = p.bias ? p.bias[16 * n + 4 * g + reg] * kLog2e : 0.f;
}
template <int KC, int NT>
__device__ __forceinline__ void hd_weights(const HdP& p, int lane, float4 (&wf)[NT][KC]) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const float4 w = reinterpret_cast<const float4*>(p.wp_fwd)[(n * KC + c) * 64 + lane];
            wf[n][c] = make_float4(w.x * kLog2e, w.y * kLog2e, w.z * kLog2e, w.w * kLog2e);
        }
}
template <int KC>
__device__ __forceinline__ void hd_prologue(const HdP& p, int g, float4 (&psc)[KC], float4 (&psf)[KC]) {
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        psc[c] = p.ps ? *reinterpret_cast<const float4*>(p.ps + 16 * c + 4 * g) : make_float4(1.f, 1.f, 1.f, 1.f);
        psf[c] = p.ps ? *reinterpret_cast<const float4*>(p.pt + 16 * c + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int KC, int NT, typename T = float>     // T: storage type of x / dx (da_bf16 = bf16 activation storage, common.h)
__global__ void __launch_bounds__(256) head_dice_fwd_kernel(HdP p) {
    __shared__ double sred[4][3][NT * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int n_s = blockIdx.y;
    const T* xs = reinterpret_cast<const T*>(p.x) + (long long)n_s * p.V * p.K;
    const long long lbase = (long long)n_s * p.V;
    float4 wf[NT][KC]; float bv[NT][4]; float4 psc[KC], psf[KC];
    hd_bias<NT>(p, g, bv);
    hd_weights<KC, NT>(p, lane, wf);
    hd_prologue<KC>(p, g, psc, psf);
    f2 sI[NT][2], sS[NT][2], sT[NT][2];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int h = 0; h < 2; ++h) sI[n][h] = sS[n][h] = sT[n][h] = (f2){0.f, 0.f};
    const long long nchunks = (p.V + 255) / 256;
    float4 an[MT][KC]; int reln[MT];
    if ((long long)blockIdx.x < nchunks) { hd_load<KC, T>(p, xs, (long long)blockIdx.x * 256 + wave * 64, i, g, an); hd_labels(p, lbase, (long long)blockIdx.x * 256 + wave * 64, i, g, reln); }
    for (long long cb = blockIdx.x; cb < nchunks; cb += gridDim.x) {
        const long long vbase = cb * 256 + wave * 64;
        float4 a[MT][KC]; f32x4 acc[MT][NT]; int rel[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            rel[t] = hd_rel(p, reln[t], vbase + t * 16 + i, g);
#pragma unroll
            for (int c = 0; c < KC; ++c) a[t][c] = an[t][c];
        }
        { const long long vn = min(cb + (long long)gridDim.x, nchunks - 1) * 256 + wave * 64; hd_load<KC, T>(p, xs, vn, i, g, an); hd_labels(p, lbase, vn, i, g, reln); }      // (the last iteration re-reads a chunk)
        hd_probs<KC, NT, T>(p, vbase, i, g, psc, psf, wf, bv, a, acc);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f2 mk[NT][2];
            hd_mask<NT>(rel[t], mk);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f2 lo = hd_lo(acc[t][n]), hi = hd_hi(acc[t][n]);
                sS[n][0] += lo; sS[n][1] += hi;
                sI[n][0] += lo * mk[n][0]; sI[n][1] += hi * mk[n][1];
                sT[n][0] += mk[n][0]; sT[n][1] += mk[n][1];
            }
        }
    }
    // per-lane fp32 sums cover at most a few hundred voxels; everything past them is double: the 16 voxel lanes, then the four waves
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            double a0 = (double)sI[n][reg >> 1][reg & 1], a1 = (double)sS[n][reg >> 1][reg & 1], a2 = (double)sT[n][reg >> 1][reg & 1];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
            if (i == 0) { const int c = 16 * n + 4 * g + reg; sred[wave][0][c] = a0; sred[wave][1][c] = a1; sred[wave][2][c] = a2; }
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 3 * p.C; idx += blockDim.x) {
        const int k = idx / p.C, c = idx % p.C;
        p.partial[(((size_t)n_s * gridDim.x + blockIdx.x) * 3 + k) * p.C + c] = (sred[0][k][c] + sred[1][k][c]) + (sred[2][k][c] + sred[3][k][c]);
    }
}

template <int KC, int NT, typename T = float>
__global__ void __launch_bounds__(256) head_dice_bwd_kernel(HdP p) {
    constexpr int K = KC * 16, C = NT * 16, LDX = K + 4, LDD = C + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    float* xl = lds + wave * (64 * LDX + 64 * LDD);          // this wave's activated-input tile [64 voxels][LDX]
    float* dl = xl + 64 * LDX;                               // this wave's d loss / d logits tile [64 voxels][LDD]
    float4 wf[NT][KC]; float bv[NT][4]; float4 psc[KC], psf[KC];
    hd_bias<NT>(p, g, bv);
    hd_weights<KC, NT>(p, lane, wf);
    hd_prologue<KC>(p, g, psc, psf);
    // data gradient, A operand: wd[c][n][reg] (lane (i, g)) = W[input channel 16c + i][class 16n + 4g + reg]
    float wd[KC][NT][4];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) wd[c][n][reg] = p.wp_bwd[(size_t)(16 * c + i) * C + 16 * n + 4 * g + reg];
    const float gl = p.dloss[0];
    f32x4 wacc[KC][NT];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int n = 0; n < NT; ++n) wacc[c][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f2 db[NT][2];
#pragma unroll
    for (int n = 0; n < NT; ++n) db[n][0] = db[n][1] = (f2){0.f, 0.f};
    // BatchNorm-backward sums of x's producer (p.bst; KC == 1): this lane's four input channels 4g .. 4g + 3 over its voxels
    const bool bst = KC == 1 && p.bst != nullptr;
    float4 bsc = make_float4(0.f, 0.f, 0.f, 0.f), bsf = bsc, bmu = bsc;
    float bs1[4] = {0.f, 0.f, 0.f, 0.f}, bs2[4] = {0.f, 0.f, 0.f, 0.f};
    if (bst) { bsc = *reinterpret_cast<const float4*>(p.ps + 4 * g); bsf = *reinterpret_cast<const float4*>(p.pt + 4 * g); bmu = *reinterpret_cast<const float4*>(p.pmean + 4 * g); }
    const long long chunks_per_sample = (p.V + 255) / 256, nchunks = chunks_per_sample * p.N;
    float4 an[MT][KC]; int reln[MT];
    auto fetch = [&](long long cb) {                           // the input quads and labels of chunk cb, one iteration ahead of their use
        const int n_s = (int)(cb / chunks_per_sample);
        const long long vbase = (cb - (long long)n_s * chunks_per_sample) * 256 + wave * 64;
        hd_load<KC, T>(p, reinterpret_cast<const T*>(p.x) + (long long)n_s * p.V * p.K, vbase, i, g, an);
        hd_labels(p, (long long)n_s * p.V, vbase, i, g, reln);
    };
    if ((long long)blockIdx.x < nchunks) fetch(blockIdx.x);
    for (long long cb = blockIdx.x; cb < nchunks; cb += gridDim.x) {
        const int n_s = (int)(cb / chunks_per_sample);
        const long long vbase = (cb - (long long)n_s * chunks_per_sample) * 256 + wave * 64;
        const T* xs = reinterpret_cast<const T*>(p.x) + (long long)n_s * p.V * p.K;
        T* dxs = reinterpret_cast<T*>(p.dx) + (long long)n_s * p.V * p.K;
        const float* c0 = p.coef + (size_t)n_s * p.C;                  // coef[0][n][c]
        const float* c1 = p.coef + (size_t)(p.N + n_s) * p.C;          // coef[1][n][c]
        float4 a[MT][KC]; f32x4 acc[MT][NT]; int rel[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            rel[t] = hd_rel(p, reln[t], vbase + t * 16 + i, g);
#pragma unroll
            for (int c = 0; c < KC; ++c) a[t][c] = an[t][c];
        }
        fetch(min(cb + (long long)gridDim.x, nchunks - 1));      // (the last iteration re-reads a chunk)
        hd_probs<KC, NT, T>(p, vbase, i, g, psc, psf, wf, bv, a, acc);
        // activated input tile -> LDS [voxel][cin] (A operand of the weight-gradient GEMM, read transposed)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int c = 0; c < KC; ++c) *reinterpret_cast<float4*>(xl + (t * 16 + i) * LDX + 16 * c + 4 * g) = a[t][c];
        f2 k0[NT][2], k1[NT][2];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                k0[n][h] = (f2){c0[16 * n + 4 * g + 2 * h], c0[16 * n + 4 * g + 2 * h + 1]};
                k1[n][h] = (f2){c1[16 * n + 4 * g + 2 * h], c1[16 * n + 4 * g + 2 * h + 1]};
            }
        // d loss / d logits = gl * p * (g - sum_c g p),  g = coef0 [label == c] + coef1   (Dice backward through the softmax Jacobian)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f2 mk[NT][2], gg[NT][2], pr[NT][2], dot2 = (f2){0.f, 0.f};
            hd_mask<NT>(rel[t], mk);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                pr[n][0] = hd_lo(acc[t][n]); pr[n][1] = hd_hi(acc[t][n]);
#pragma unroll
                for (int h = 0; h < 2; ++h) { gg[n][h] = k0[n][h] * mk[n][h] + k1[n][h]; dot2 += gg[n][h] * pr[n][h]; }
            }
            const f2 dot = hd_bc(da_rows_sum(dot2.x + dot2.y)), gl2 = hd_bc(gl);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f2 d0 = (gl2 * pr[n][0]) * (gg[n][0] - dot), d1 = (gl2 * pr[n][1]) * (gg[n][1] - dot);       // voxels past the end: p = 0 -> 0
                db[n][0] += d0; db[n][1] += d1;
                acc[t][n] = (f32x4){d0.x, d0.y, d1.x, d1.y};
                *reinterpret_cast<float4*>(dl + (t * 16 + i) * LDD + 16 * n + 4 * g) = make_float4(d0.x, d0.y, d1.x, d1.y);
            }
        }
        // data gradient: dx^T[k][voxel] = sum_c W[k][c] dl[c][voxel]; the dl registers are the B operand as they stand
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            f32x4 dacc[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) dacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                    for (int c = 0; c < KC; ++c) dacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wd[c][n][reg], acc[t][n][reg], dacc[c], 0, 0, 0);
            const long long vox = vbase + t * 16 + i;
            if (vox < p.V) {
#pragma unroll
                for (int c = 0; c < KC; ++c)          // lane (voxel i, group g) holds input channels 16c + 4g .. + 3
                    da_stq_nt(dxs, (vox * p.K + 16 * c + 4 * g) >> 2, make_float4(dacc[c][0], dacc[c][1], dacc[c][2], dacc[c][3]));
                if (bst) {                            // the raw x of this voxel again (an L1 / L2 hit: this wave loaded it for the logits a moment ago)
                    const float4 xr = da_ldq(xs, (vox * p.K + 4 * g) >> 2);
                    const float xv[4] = {xr.x, xr.y, xr.z, xr.w}, scv[4] = {bsc.x, bsc.y, bsc.z, bsc.w}, sfv[4] = {bsf.x, bsf.y, bsf.z, bsf.w}, muv[4] = {bmu.x, bmu.y, bmu.z, bmu.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dz = dacc[0][j] * da_act_grad(xv[j] * scv[j] + sfv[j], p.pslope >= 1.f ? -1.f : p.pslope);
                        bs1[j] += dz; bs2[j] += dz * (xv[j] - muv[j]);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();       // xl / dl are this wave's own tiles and a wave's LDS accesses execute in order: no workgroup barrier
        // weight gradient: dW[k][c] += sum_voxels xa[voxel][k] dl[voxel][c]   (K dimension = the 64 voxels of this wave's chunk)
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            float av[KC], bw[NT];
#pragma unroll
            for (int c = 0; c < KC; ++c) av[c] = xl[(4 * s + g) * LDX + 16 * c + i];
#pragma unroll
            for (int n = 0; n < NT; ++n) bw[n] = dl[(4 * s + g) * LDD + 16 * n + i];
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int n = 0; n < NT; ++n) wacc[c][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bw[n], wacc[c][n], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // per-workgroup partials: dW [K][C] then dbias [C]; the four waves are folded through LDS in a fixed order
    __syncthreads();
    float* red = lds;                                         // [K*C + C]
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int idx = (16 * c + 4 * g + reg) * C + 16 * n + i;         // row (input channel) = 4g + reg, col (class) = i
                        red[idx] = (w == 0 ? 0.f : red[idx]) + wacc[c][n][reg];
                    }
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    float d = db[n][reg >> 1][reg & 1];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) d += __shfl_xor(d, o);
                    const int cidx = K * C + 16 * n + 4 * g + reg;
                    if (i == 0) red[cidx] = (w == 0 ? 0.f : red[cidx]) + d;
                }
        }
        __syncthreads();
    }
    float* part = p.wpartial + (size_t)blockIdx.x * (K * C + C);
    for (int idx = threadIdx.x; idx < K * C + C; idx += blockDim.x) part[idx] = red[idx];
    if (bst) {                                                // per-lane fp32 sums (<= a few hundred voxels) -> doubles over the 16 voxel lanes -> the four waves
        __syncthreads();
        double* dred = reinterpret_cast<double*>(lds);        // [wave][2][16]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double a0 = (double)bs1[j], a1 = (double)bs2[j];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); }
            if (i == 0) { dred[(wave * 2 + 0) * 16 + 4 * g + j] = a0; dred[(wave * 2 + 1) * 16 + 4 * g + j] = a1; }
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const int k = threadIdx.x >> 4, c = threadIdx.x & 15;
            p.bst[((size_t)blockIdx.x * 2 + k) * 16 + c] = (dred[(0 * 2 + k) * 16 + c] + dred[(1 * 2 + k) * 16 + c]) + (dred[(2 * 2 + k) * 16 + c] + dred[(3 * 2 + k) * 16 + c]);
        }
    }
}

static bool hd_shape_ok(int K, int C) { return (K == 16 || K == 64) && (C == 16 || C == 32); }

}  // namespace

extern "C" size_t da_head_dice_ws_bytes(int N, long long V, int Cin, int C) {
    (void)V;
    const size_t pack = da_align((size_t)2 * Cin * C * sizeof(float));
    const size_t fwd = da_align((size_t)N * kHdBlocks * 3 * C * sizeof(double)) + da_align((size_t)3 * N * C * sizeof(float));
    const size_t bwd = da_align((size_t)kHdBlocks * (Cin * C + C) * sizeof(float)) + da_align((size_t)(Cin * C + C) * sizeof(float));
    return pack + (fwd > bwd ? fwd : bwd);
}

template <int KC, int NT, typename T>
static int hd_launch_fwd(const HdP& p, int nblocks, hipStream_t st) {
    hipLaunchKernelGGL((head_dice_fwd_kernel<KC, NT, T>), dim3(nblocks, p.N), dim3(256), 0, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}
template <int KC, int NT, typename T>
static int hd_launch_bwd(const HdP& p, int nblocks, hipStream_t st) {
    const size_t shm = (size_t)4 * (64 * (KC * 16 + 4) + 64 * (NT * 16 + 4)) * sizeof(float);
    auto kern = head_dice_bwd_kernel<KC, NT, T>;
    static bool attr_set = false;
    if (!attr_set && shm > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), shm, st, p);
    DA_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int head_dice_fwd_t(const float* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                           const float* w_io, const float* bias, const void* labels, int label_bytes,
                           int N, long long V, int Cin, int C, int weight_type, int no_bg, float eps,
                           float* loss, float* coef, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !w_io || !labels || !loss || !coef || N <= 0 || N > 64 || V <= 0 || (label_bytes != 1 && label_bytes != 8) ||
        ((pro_scale == nullptr) != (pro_shift == nullptr))) return DA_ERR_BADARG;
    if (!hd_shape_ok(Cin, C) || (pro_scale && pro_slope >= 1.f)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_head_dice_ws_bytes(N, V, Cin, C)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    float* wp = (float*)ws;
    hipLaunchKernelGGL(hd_pack_kernel, dim3(da_grid(Cin * C, 256, 64)), dim3(256), 0, st, w_io, wp, Cin, C, 0);
    DA_LAUNCH_CHECK();
    char* rest = (char*)ws + da_align((size_t)2 * Cin * C * sizeof(float));
    double* partial = (double*)rest;
    float* isc = (float*)(rest + da_align((size_t)N * kHdBlocks * 3 * C * sizeof(double)));
    int nblocks = (int)da_cdiv(V, 256 * 2); if (nblocks > kHdBlocks) nblocks = kHdBlocks; if (nblocks < 1) nblocks = 1;
    HdP p;
    p.x = x; p.ps = pro_scale; p.pt = pro_shift; p.pslope = pro_scale ? (pro_slope < 0.f ? 1.f : pro_slope) : 1.f;
    p.wp_fwd = wp; p.wp_bwd = nullptr; p.bias = bias; p.labels = labels; p.label_bytes = label_bytes;
    p.V = V; p.N = N; p.K = Cin; p.C = C; p.partial = partial; p.coef = nullptr; p.dloss = nullptr; p.dx = nullptr; p.wpartial = nullptr; p.pmean = nullptr; p.bst = nullptr;
    int rc;
    if (Cin == 16 && C == 32) rc = hd_launch_fwd<1, 2, T>(p, nblocks, st);
    else if (Cin == 16 && C == 16) rc = hd_launch_fwd<1, 1, T>(p, nblocks, st);
    else if (Cin == 64 && C == 32) rc = hd_launch_fwd<4, 2, T>(p, nblocks, st);
    else rc = hd_launch_fwd<4, 1, T>(p, nblocks, st);
    if (rc) return rc;
    return da_dice_finish(partial, nblocks, N, C, weight_type, no_bg, eps, loss, coef, isc, st);
}
extern "C" int da_head_dice_fwd(const float* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                                const float* w_io, const float* bias, const void* labels, int label_bytes,
                                int N, long long V, int Cin, int C, int weight_type, int no_bg, float eps,
                                float* loss, float* coef, void* ws, size_t ws_bytes, void* stream) {
    return head_dice_fwd_t<float>(x, pro_scale, pro_shift, pro_slope, w_io, bias, labels, label_bytes, N, V, Cin, C, weight_type, no_bg, eps, loss, coef, ws, ws_bytes, stream);
}
extern "C" int da_head_dice_fwd_bf16(const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                                     const float* w_io, const float* bias, const void* labels, int label_bytes,
                                     int N, long long V, int Cin, int C, int weight_type, int no_bg, float eps,
                                     float* loss, float* coef, void* ws, size_t ws_bytes, void* stream) {
    return head_dice_fwd_t<da_bf16>((const float*)x, pro_scale, pro_shift, pro_slope, w_io, bias, labels, label_bytes, N, V, Cin, C, weight_type, no_bg, eps, loss, coef, ws, ws_bytes, stream);
}

template <typename T>
static int head_dice_bwd_t(const float* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                           const float* w_io, const float* bias, const void* labels, int label_bytes,
                           const float* coef, const float* dloss, float* dx, float* dw_io, float* dbias,
                           int N, long long V, int Cin, int C, void* ws, size_t ws_bytes, void* stream,
                           const float* pro_mean = nullptr, double* bst = nullptr, int bst_cap = 0, int* bst_n = nullptr) {
    if (bst_n) *bst_n = 0;
    if (!x || !w_io || !labels || !coef || !dloss || !dx || !dw_io || N <= 0 || V <= 0 || (label_bytes != 1 && label_bytes != 8) ||
        ((pro_scale == nullptr) != (pro_shift == nullptr))) return DA_ERR_BADARG;
    if (!hd_shape_ok(Cin, C) || (pro_scale && pro_slope >= 1.f)) return DA_ERR_UNSUPPORTED;
    if (ws_bytes < da_head_dice_ws_bytes(N, V, Cin, C)) return DA_ERR_WS_SMALL;
    hipStream_t st = da_stream(stream);
    float* wpf = (float*)ws;
    const float* wpb = w_io;                 // the data-gradient A fragments are read straight from w_io [Cin][C]
    hipLaunchKernelGGL(hd_pack_kernel, dim3(da_grid(Cin * C, 256, 64)), dim3(256), 0, st, w_io, wpf, Cin, C, 0);
    DA_LAUNCH_CHECK();
    char* rest = (char*)ws + da_align((size_t)2 * Cin * C * sizeof(float));
    float* wpartial = (float*)rest;
    const int O = Cin * C + C;
    float* folded = (float*)(rest + da_align((size_t)kHdBlocks * O * sizeof(float)));
    const long long nchunks = da_cdiv(V, 256) * N;
    int nblocks = (int)(nchunks < kHdBlocks ? nchunks : kHdBlocks); if (nblocks < 1) nblocks = 1;
    HdP p;
    p.x = x; p.ps = pro_scale; p.pt = pro_shift; p.pslope = pro_scale ? (pro_slope < 0.f ? 1.f : pro_slope) : 1.f;
    p.wp_fwd = wpf; p.wp_bwd = wpb; p.bias = bias; p.labels = labels; p.label_bytes = label_bytes;
    p.V = V; p.N = N; p.K = Cin; p.C = C; p.partial = nullptr; p.coef = coef; p.dloss = dloss; p.dx = dx; p.wpartial = wpartial;
    // the producer's BatchNorm-backward sums ride along when x carries a deferred BatchNorm (16 input channels: one channel quad per lane group)
    const bool want_bst = bst && pro_scale && pro_mean && Cin == 16 && bst_cap >= nblocks && !DaEl<T>::bf;
    p.pmean = want_bst ? pro_mean : nullptr; p.bst = want_bst ? bst : nullptr;
    if (want_bst && bst_n) *bst_n = nblocks;
    int rc;
    if (Cin == 16 && C == 32) rc = hd_launch_bwd<1, 2, T>(p, nblocks, st);
    else if (Cin == 16 && C == 16) rc = hd_launch_bwd<1, 1, T>(p, nblocks, st);
    else if (Cin == 64 && C == 32) rc = hd_launch_bwd<4, 2, T>(p, nblocks, st);
    else rc = hd_launch_bwd<4, 1, T>(p, nblocks, st);
    if (rc) return rc;
    rc = da_reduce_partials(wpartial, nblocks, O, folded, st);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(dw_io, folded, (size_t)Cin * C * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    if (dbias) { e = hipMemcpyAsync(dbias, folded + (size_t)Cin * C, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st); if (e != hipSuccess) return (int)e; }
    return 0;
}
extern "C" int da_head_dice_bwd(const float* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                                const float* w_io, const float* bias, const void* labels, int label_bytes,
                                const float* coef, const float* dloss, float* dx, float* dw_io, float* dbias,
                                int N, long long V, int Cin, int C, void* ws, size_t ws_bytes, void* stream) {
    return head_dice_bwd_t<float>(x, pro_scale, pro_shift, pro_slope, w_io, bias, labels, label_bytes, coef, dloss, dx, dw_io, dbias, N, V, Cin, C, ws, ws_bytes, stream);
}
// The same, plus the BatchNorm-backward sums of the layer that produced x (x = its raw convolution output, (pro_scale, pro_shift, pro_mean) its
// batch statistics): bst[bst_n][2][Cin] doubles = (sum dz, sum dz (x - mean)) per workgroup, dz = dx act'(x scale + shift) -- what
// da_bn_act_bwd_dbias_pre takes instead of its own reduction pass over (dx, x).  *bst_n = 0 when the shape has no such epilogue (Cin != 16, bf16 storage,
// bst_cap < 1024): the caller then runs the stand-alone pass.  autograd of unets.py:31-32 on the last decoder block.
extern "C" int da_head_dice_bwd_bst(const float* x, const float* pro_scale, const float* pro_shift, float pro_slope, const float* pro_mean,
                                    const float* w_io, const float* bias, const void* labels, int label_bytes,
                                    const float* coef, const float* dloss, float* dx, float* dw_io, float* dbias,
                                    int N, long long V, int Cin, int C, double* bst, int bst_cap, int* bst_n, void* ws, size_t ws_bytes, void* stream) {
    return head_dice_bwd_t<float>(x, pro_scale, pro_shift, pro_slope, w_io, bias, labels, label_bytes, coef, dloss, dx, dw_io, dbias, N, V, Cin, C, ws, ws_bytes, stream,
                                  pro_mean, bst, bst_cap, bst_n);
}
extern "C" int da_head_dice_bwd_bf16(const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                                     const float* w_io, const float* bias, const void* labels, int label_bytes,
                                     const float* coef, const float* dloss, void* dx, float* dw_io, float* dbias,
                                     int N, long long V, int Cin, int C, void* ws, size_t ws_bytes, void* stream) {
    return head_dice_bwd_t<da_bf16>((const float*)x, pro_scale, pro_shift, pro_slope, w_io, bias, labels, label_bytes, coef, dloss, (float*)dx, dw_io, dbias, N, V, Cin, C, ws, ws_bytes, stream);
}
