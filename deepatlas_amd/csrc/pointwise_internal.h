// Internal entry points of pointwise_mfma.hip (MFMA channel GEMMs for 1x1 conv and 2x2x2/s2 transposed conv).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

bool da_pw_supported(int K, int N);
size_t da_pw_packed_bytes(int ntaps, int K, int N);
// SCATTER (gather == 0): out[map(v,t)][N] = bias + A[v][K] * B_t ; GATHER: out[v][N] = sum_t A[map(v,t)][K] * B_t
int da_pw_gemm(const float* a, const float* w, int transposed, const float* bias, float* out,
               long long M, int D, int H, int W, int K, int N, int ntaps, int up, int gather,
               void* ws, size_t ws_bytes, hipStream_t st, double* stats_partial = nullptr,   // stats_partial: [cdiv(M,256)][2][N] BatchNorm sums (scatter form)
               const float* pro_scale = nullptr, const float* pro_shift = nullptr, float pro_slope = -1.f,   // input prologue: act(a * scale + shift) is consumed
               int a_bf16 = 0, int out_bf16 = 0);     // bf16 activation storage: `a` / `out` point to bf16 tensors
size_t da_pw_wgrad_ws_bytes(long long M, int ntaps, int Cin, int Cout);
int da_pw_wgrad(const float* in, const float* dy, float* dw, long long M, int D, int H, int W, int Cin, int Cout,
                int ntaps, int up, void* ws, size_t ws_bytes, hipStream_t st,
                const float* pro_scale = nullptr, const float* pro_shift = nullptr, float pro_slope = -1.f, int in_bf16 = 0, int dy_bf16 = 0);

// norm_act.hip: the sums of a training-mode BatchNorm + activation backward (dgamma, dbeta, cm = (mean dz, mean dz xhat)); cm lies inside ws
int da_bn_bwd_sums(const float* dy, const float* x, const float* mean, const float* rstd, const float* scale, const float* shift, float act_slope,
                   long long M, int C, float* dgamma, float* dbeta, const float** cm_out, void* ws, size_t ws_bytes, hipStream_t st, const double* pre, int pre_n);
// pointwise_mfma.hip: fused BatchNorm-backward apply + transposed-conv (k2 s2) data gradient + weight gradient (da_deconv_k2s2_bn_bwd, deconv.hip)
bool da_deconv_bn_bwd_supported(int Cin, int Cout);
size_t da_deconv_bn_bwd_ws_bytes(long long M, int Cin, int Cout);
int da_deconv_bn_bwd(const float* gout, const float* y, const float* mean, const float* rstd, const float* scale, const float* shift, const float* cm, float slope,
                     const float* in, const float* w_tio, float* dx, float* dw_tio, float* dbias,
                     long long M, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, hipStream_t st);
