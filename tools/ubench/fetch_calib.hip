// FETCH_SIZE calibration on known byte counts (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern before
// trusting an absolute").  Streams a 1 GiB buffer (>> 256 MiB Infinity Cache + 32 MiB L2) once with raw_buffer_load_b128, in the two
// request shapes the 3x3x3 conv kernels' halo staging produces:
//   stream_b128_contig : 16 B per lane, a wave reads 1 KiB contiguous            (staging a 16-channel tensor: a voxel = one 64-B run,
//                                                                                 consecutive voxels contiguous)
//   stream_b128_half   : 64-B runs at a 128-B stride (4 lanes per run), i.e. only the first 16 channels of a 32-channel tensor -- the
//                        other half of every 128-B line is NOT requested by this kernel
//   stream_b128_run32_s64 / _s128 : 32-B runs (2 lanes per run) at a 64-B / 128-B stride -- the split-mode kernels' 8-channel chunks of a
//                        16- / 32-channel tensor (the rest of each line is requested by the NEXT chunk's item, normally an L2 hit)
// Build + run under rocprofv3 --pmc FETCH_SIZE (tools/pmc_conv.sh); prints the bytes each kernel requested.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

__global__ void stream_b128_contig(const float* __restrict__ src, float* __restrict__ sink, unsigned long long nvec) {
    // per-workgroup descriptors of <= 1 GiB so that offsets stay 32-bit, as in the conv kernels (one descriptor per sample)
    const __amdgpu_buffer_rsrc_t r = rsrc(src, 0x40000000u);
    float acc = 0.f;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(i * 16ull), 0, 0));
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1234.5f) sink[0] = acc;
}

__global__ void stream_b128_half(const float* __restrict__ src, float* __restrict__ sink, unsigned long long nvec) {
    const __amdgpu_buffer_rsrc_t r = rsrc(src, 0x40000000u);
    float acc = 0.f;
    // vector i = (voxel v = i / 4, quad q = i % 4): byte offset v * 128 + q * 16 -> the first 64 B of every 128-B voxel row
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long v = i >> 2, q = i & 3;
        const float4 x = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(v * 128ull + q * 16ull), 0, 0));
        acc += x.x + x.y + x.z + x.w;
    }
    if (acc == 1234.5f) sink[0] = acc;
}

template <int STRIDE>
__global__ void stream_b128_run32(const float* __restrict__ src, float* __restrict__ sink, unsigned long long nvec) {
    const __amdgpu_buffer_rsrc_t r = rsrc(src, 0x40000000u);
    float acc = 0.f;
    // vector i = (voxel v = i / 2, quad q = i % 2): byte offset v * STRIDE + q * 16 -> the first 32 B of every STRIDE-byte voxel row
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long v = i >> 1, q = i & 1;
        const float4 x = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(v * (unsigned long long)STRIDE + q * 16ull), 0, 0));
        acc += x.x + x.y + x.z + x.w;
    }
    if (acc == 1234.5f) sink[0] = acc;
}

int main() {
    const size_t bytes = 1ull << 30;
    float *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_b128_contig, dim3(8192), dim3(256), 0, 0, buf, sink, (unsigned long long)(bytes / 16));
        hipLaunchKernelGGL(stream_b128_half, dim3(8192), dim3(256), 0, 0, buf, sink, (unsigned long long)(bytes / 32));
        hipLaunchKernelGGL(stream_b128_run32<64>, dim3(8192), dim3(256), 0, 0, buf, sink, (unsigned long long)(bytes / 32));
        hipLaunchKernelGGL(stream_b128_run32<128>, dim3(8192), dim3(256), 0, 0, buf, sink, (unsigned long long)(bytes / 64));
    }
    hipDeviceSynchronize();
    printf("stream_b128_contig requested_bytes %llu\nstream_b128_half requested_bytes %llu\n", (unsigned long long)bytes, (unsigned long long)(bytes / 2));
    printf("stream_b128_run32<64> requested_bytes %llu\nstream_b128_run32<128> requested_bytes %llu\n", (unsigned long long)(bytes / 2), (unsigned long long)(bytes / 4));
    return 0;
}
