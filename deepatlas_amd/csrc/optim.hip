// Adam (torch.optim.Adam defaults, models/segmentation.py:91) over one flat parameter bucket, weight layout
// transforms between the reference's state_dict layouts and the kernels' tap-major layouts, library info.
#include "common.h"
#include <string.h>

namespace {

// one element of the update; contraction off so that the two kernels below (scalars as launch arguments / scalars from device memory)
// round identically whatever the compiler would otherwise fuse
__device__ __forceinline__ void adam_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                            long long i, float lr_over_bc1, float beta1, float beta2, float eps, float inv_sqrt_bc2, float gscale) {
#pragma clang fp contract(off)
    const float gi = g[i] * gscale;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;       // sqrt(v)/sqrt(bc2) + eps
    p[i] = p[i] - lr_over_bc1 * (mi / denom);
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long long n, float lr_over_bc1, float beta1, float beta2, float eps, float inv_sqrt_bc2, float gscale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        adam_update(p, g, m, v, i, lr_over_bc1, beta1, beta2, eps, inv_sqrt_bc2, gscale);
}

// The same update with every per-step scalar read from DEVICE memory, so that the launch can sit in a captured HIP graph and be
// replayed: state6 = {lr / bc1, beta1, beta2, eps, 1 / sqrt(bc2), grad_scale}, computed on the HOST for the step about to run with exactly
// the expressions of da_adam_step (bit-identical updates) and copied over before each replay.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                long long n, const float* __restrict__ state) {
    const float lr_over_bc1 = state[0], beta1 = state[1], beta2 = state[2], eps = state[3], inv_sqrt_bc2 = state[4], gscale = state[5];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        adam_update(p, g, m, v, i, lr_over_bc1, beta1, beta2, eps, inv_sqrt_bc2, gscale);
}

// generic 3-axis permutation of a [A][B][K] tensor into [K][P][Q]; mode selects which of (A,B) is Cin
__global__ void w_to_tio_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int B, int K, int a_is_cout) {
    const long long total = (long long)A * B * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K); const int b = (int)((i / K) % B); const int a = (int)(i / ((long long)K * B));
        // src[a][b][k]; dst[k][ci][co]   (flags bit 0: a is cout; bit 1: taps flipped, k -> K-1-k)
        const bool aco = (a_is_cout & 1) != 0;
        const int Cin = aco ? B : A, Cout = aco ? A : B;
        const int ci = aco ? b : a, co = aco ? a : b;
        const int kk = (a_is_cout & 2) ? K - 1 - k : k;
        dst[((size_t)kk * Cin + ci) * Cout + co] = src[i];
    }
}
__global__ void tio_to_w_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int B, int K, int a_is_cout) {
    const long long total = (long long)A * B * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K); const int b = (int)((i / K) % B); const int a = (int)(i / ((long long)K * B));
        const bool aco = (a_is_cout & 1) != 0;
        const int Cin = aco ? B : A, Cout = aco ? A : B;
        const int ci = aco ? b : a, co = aco ? a : b;
        const int kk = (a_is_cout & 2) ? K - 1 - k : k;
        const float v = src[((size_t)kk * Cin + ci) * Cout + co];
        dst[i] = (a_is_cout & 4) ? dst[i] + v : v;             // bit 2: accumulate (gradient straight into the optimiser's bucket)
    }
}

}  // namespace

// out[o] = sum over partial sets b of partial[b][o], in double, in a fixed order (deterministic).  16 outputs x 16 set-groups per
// block: a thread walks nparts / 16 sets (four loads in flight), the 16 group sums of an output meet in LDS.  The partial sets of
// a weight gradient are a few MB and L2-resident; what matters is the length of the per-thread dependent chain, not bandwidth.
__global__ void __launch_bounds__(256) da_reduce_partials_kernel(const float* __restrict__ partial, int nparts, int O, float* __restrict__ out) {
    __shared__ double sh[16][17];
    const int ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + ol;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (o < O) {
        int b = sl;
        for (; b + 48 < nparts; b += 64) {
            s0 += (double)partial[(size_t)b * O + o]; s1 += (double)partial[(size_t)(b + 16) * O + o];
            s2 += (double)partial[(size_t)(b + 32) * O + o]; s3 += (double)partial[(size_t)(b + 48) * O + o];
        }
        for (; b < nparts; b += 16) s0 += (double)partial[(size_t)b * O + o];
    }
    sh[sl][ol] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && o < O) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][ol];
        out[o] = (float)t;
    }
}

int da_reduce_partials(const float* partial, int nparts, int O, float* out, hipStream_t st) {
    hipLaunchKernelGGL(da_reduce_partials_kernel, dim3((O + 15) / 16), dim3(256), 0, st, partial, nparts, O, out);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_version(void) { return 100; }

extern "C" int da_device_info(int* cu_count, int* wave_size, size_t* hbm_bytes, char* arch, int arch_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return 0;
}

extern "C" int da_adam_step(float* p, const float* g, float* m, float* v, long long n,
                            float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return DA_ERR_BADARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(da_grid(n, 256)), dim3(256), 0, da_stream(stream), p, g, m, v, n,
                       (float)((double)lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
    DA_LAUNCH_CHECK();
    return 0;
}

extern "C" int da_adam_host_state(float lr, float beta1, float beta2, float eps, int step, float grad_scale, float* state6_host) {
    if (!state6_host || step < 1) return DA_ERR_BADARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    state6_host[0] = (float)((double)lr / bc1); state6_host[1] = beta1; state6_host[2] = beta2; state6_host[3] = eps;
    state6_host[4] = (float)(1.0 / sqrt(bc2)); state6_host[5] = grad_scale;
    return 0;
}

extern "C" int da_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float* state6, void* stream) {
    if (!p || !g || !m || !v || !state6 || n <= 0) return DA_ERR_BADARG;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(da_grid(n, 256)), dim3(256), 0, da_stream(stream), p, g, m, v, n, (const float*)state6);
    DA_LAUNCH_CHECK();
    return 0;
}

#define DA_W_LAUNCH(KERN, A, B, K, FLAG)                                                                              \
    do {                                                                                                              \
        if (!src || !dst || (A) <= 0 || (B) <= 0 || (K) <= 0) return DA_ERR_BADARG;                                   \
        hipLaunchKernelGGL(KERN, dim3(da_grid((long long)(A) * (B) * (K), 256, 1024)), dim3(256), 0, da_stream(stream), src, dst, A, B, K, FLAG); \
        DA_LAUNCH_CHECK();                                                                                            \
        return 0;                                                                                                     \
    } while (0)

extern "C" int da_w_oik_to_tio(const float* src, float* dst, int Cout, int Cin, int K3, void* stream) { DA_W_LAUNCH(w_to_tio_kernel, Cout, Cin, K3, 1); }
extern "C" int da_w_tio_to_oik(const float* src, float* dst, int Cout, int Cin, int K3, void* stream) { DA_W_LAUNCH(tio_to_w_kernel, Cout, Cin, K3, 1); }
extern "C" int da_w_iok_to_tio(const float* src, float* dst, int Cin, int Cout, int K3, void* stream) { DA_W_LAUNCH(w_to_tio_kernel, Cin, Cout, K3, 0); }
extern "C" int da_w_tio_to_iok(const float* src, float* dst, int Cin, int Cout, int K3, void* stream) { DA_W_LAUNCH(tio_to_w_kernel, Cin, Cout, K3, 0); }
// ConvTranspose3d(k, stride 1, padding (k-1)/2) == Conv3d with the taps flipped and the channel axes swapped
extern "C" int da_w_iok_flip_to_tio(const float* src, float* dst, int Cin, int Cout, int K3, void* stream) { DA_W_LAUNCH(w_to_tio_kernel, Cin, Cout, K3, 2); }
extern "C" int da_w_tio_to_iok_flip(const float* src, float* dst, int Cin, int Cout, int K3, void* stream) { DA_W_LAUNCH(tio_to_w_kernel, Cin, Cout, K3, 2); }
// ... and accumulating: dst += layout(src) (one launch instead of a conversion + an add; dst is a view of FlatAdam's gradient bucket)
extern "C" int da_w_tio_to_oik_acc(const float* src, float* dst, int Cout, int Cin, int K3, void* stream) { DA_W_LAUNCH(tio_to_w_kernel, Cout, Cin, K3, 5); }
extern "C" int da_w_tio_to_iok_acc(const float* src, float* dst, int Cin, int Cout, int K3, void* stream) { DA_W_LAUNCH(tio_to_w_kernel, Cin, Cout, K3, 4); }
extern "C" int da_w_tio_to_iok_flip_acc(const float* src, float* dst, int Cin, int Cout, int K3, void* stream) { DA_W_LAUNCH(tio_to_w_kernel, Cin, Cout, K3, 6); }
