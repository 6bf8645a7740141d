// Internal (non-ABI) entry points shared between conv3d.hip (dispatch + direct kernels) and
// conv3d_mfma.hip (implicit-GEMM MFMA kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

size_t da_conv3_mfma_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int stride);

// forward-shaped implicit GEMM.  in = concat(in1[C1], in2[C2]); output channels [0,Cs1) -> out1, rest -> out2.
// w_is_flipped_tr != 0: `w_tio` is the ORIGINAL layer's [27][Cout][C1] tensor and the kernel runs the data-gradient
// convolution (taps flipped, channels transposed) -- i.e. logical Cin = C1, logical Cout = `Cout`.
bool da_matrix_bf16();                 // bf16 matrix mode switch (da_set_matrix_bf16)
int da_matrix_mode();                  // 0 fp32 MFMA | 1 bf16-rounded operands | 2 two-term fp16 split (da_set_matrix_mode, split_f16.h)
// Input prologue: in1 / in2 are RAW outputs of a producer whose per-channel affine (BatchNorm scale / shift) and activation
// (slope as in da_conv3d_k3_fwd: < 0 identity, 0 ReLU, > 0 LeakyReLU) are applied while the tile is staged.  A null scale
// pointer = that input is already activated.
struct DaPro { const float* s1; const float* t1; float slope1; const float* s2; const float* t2; float slope2; };
// Stride-2 layers (conv3d_s2.hip) run as tap-masked stride-1 convolutions over the space-to-depth view of their input (s2d_cin > 0).
// fuse_in: `in1` is the ORIGINAL tensor (s2d_cin channels, D0 x H0 x W0) and the staging loads apply the view; fuse_out (data gradient):
// `out1` is the original-resolution gradient and the epilogue stores apply the inverse view -- no space_to_depth / depth_to_space copies.
struct DaS2dFuse { int D0, H0, W0, fuse_in, fuse_out; };
bool da_conv3_mfma_fwd_supported(int C1, int C2, int Cout, int stride, int Cs1 = -1, int Cs2 = 0);   // Cs1/Cs2: output split (dgrad of a concat conv)
int da_conv3_mfma_fwd(const float* in1, int C1, const float* in2, int C2, const float* w_tio, int w_is_flipped_tr,
                      const float* bias, float* out1, int Cs1, float* out2, int Cs2,
                      int N, int D, int H, int W, int Cout, int stride, float slope,
                      void* ws, size_t ws_bytes, hipStream_t st, int s2d_cin = 0, double* stats_partial = nullptr, int* stats_nparts = nullptr,
                      const DaPro* pro = nullptr, const DaS2dFuse* s2f = nullptr, int act_bf16 = 0);   // act_bf16: in1 / in2 / out1 / out2 are bf16 tensors (bf16 matrix mode only, else DA_ERR_UNSUPPORTED)

bool da_conv3_mfma_wgrad_supported(int C1, int C2, int Cout, int stride);
int da_conv3_mfma_wgrad(const float* in1, int C1, const float* in2, int C2, const float* dy, float* dw_tio,
                        int N, int D, int H, int W, int Cout, int stride, void* ws, size_t ws_bytes, hipStream_t st, int s2d_cin = 0,
                        const DaPro* pro = nullptr, const DaS2dFuse* s2f = nullptr, int act_bf16 = 0);   // act_bf16: in1 / in2 / dy are bf16 tensors

int da_conv3_direct_fwd(const float* in1, int C1, const float* in2, int C2, const float* w, const float* bias,
                        float* out1, int Cs1, float* out2, int Cs2,
                        int N, int D, int H, int W, int Cout, int stride, float slope, hipStream_t st);

// stride-2 via space-to-depth + tap-masked stride-1 MFMA kernels (conv3d_s2.hip)
bool da_conv3_s2_supported(int C1, int C2, int Cout);
size_t da_conv3_s2_ws_bytes(int N, int D, int H, int W, int Cin, int Cout);
int da_conv3_s2_fwd(const float* in, int Cin, const float* w_tio, const float* bias, float* out,
                    int N, int D, int H, int W, int Cout, float slope, void* ws, size_t ws_bytes, hipStream_t st, int act_bf16 = 0);
int da_conv3_s2_dgrad(const float* dy, const float* w_tio, float* dx, int Cin, int N, int D, int H, int W, int Cout,
                      void* ws, size_t ws_bytes, hipStream_t st, int act_bf16 = 0);
int da_conv3_s2_wgrad(const float* in, int Cin, const float* dy, float* dw_tio, int N, int D, int H, int W, int Cout,
                      void* ws, size_t ws_bytes, hipStream_t st, int act_bf16 = 0);      // act_bf16: activation / gradient tensors are bf16 (common.h)

// native stride-2 kernels in split matrix mode (conv3d_s2n.hip): one halo tile serves all 27 taps
bool da_conv3_s2n_supported(int Cin, int Cout, int N, int D, int H, int W);
size_t da_conv3_s2n_ws_bytes(int N, int D, int H, int W, int Cin, int Cout);
int da_conv3_s2n_fwd(const float* in, int Cin, const float* w_tio, const float* bias, float* out,
                     int N, int D, int H, int W, int Cout, float slope, void* ws, size_t ws_bytes, hipStream_t st);
int da_conv3_s2n_dgrad(const float* dy, const float* w_tio, float* dx, int Cin, int N, int D, int H, int W, int Cout,
                       void* ws, size_t ws_bytes, hipStream_t st);
int da_conv3_s2n_wgrad(const float* in, int Cin, const float* dy, float* dw_tio, int N, int D, int H, int W, int Cout,
                       void* ws, size_t ws_bytes, hipStream_t st);

// the flow convolution (<= 3 output channels) and its data gradient on the matrix cores, split matrix mode (conv3d_flowmm.hip)
bool da_conv3_flowmm_supported(int C1, int C2, int Cout, int N, int D, int H, int W);
size_t da_conv3_flowmm_ws_bytes();
int da_conv3_flowmm_fwd(const float* in1, int C1, const float* in2, int C2, const float* w_tio, const float* bias, float* out,
                        int N, int D, int H, int W, int Cout, float slope, void* ws, size_t ws_bytes, hipStream_t st);
int da_conv3_flowmm_dgrad(const float* dy, const float* w_tio, float* dx1, int C1, float* dx2, int C2,
                          int N, int D, int H, int W, int Cout, void* ws, size_t ws_bytes, hipStream_t st);

// thin convs (Cout <= 4 or Cin <= 4) on the VALU from an LDS halo tile; flip_tr: `w` is the original layer's [27][Cout][Cin]
// tensor and the data-gradient convolution is computed
bool da_conv3_thin_supported(int C1, int C2, int Cout, int stride);
int da_conv3_thin_fwd(const float* in1, int C1, const float* in2, int C2, const float* w, int flip_tr, const float* bias,
                      float* out1, int Cs1, float* out2, int Cs2, int N, int D, int H, int W, int Cout, float slope,
                      void* ws, size_t ws_bytes, hipStream_t st, int in_bf16 = 0, int out_bf16 = 0);   // bf16 activation storage: inputs / outputs are bf16 tensors

// weight gradient of a conv with <= 3 output channels and <= 16 + 16 input channels (the 24 -> 3 flow conv): LDS-tiled MFMA GEMM
// with M = (tap, cout) (conv3d_flow.hip)
bool da_conv3_flow_wgrad_supported(int C1, int C2, int Cout, int stride);
size_t da_conv3_flow_wgrad_ws_bytes(int Cin, int Cout);
int da_conv3_flow_wgrad(const float* in1, int C1, const float* in2, int C2, const float* dy, float* dw_tio,
                        int N, int D, int H, int W, int Cout, void* ws, size_t ws_bytes, hipStream_t st, int in_bf16 = 0);      // in_bf16: in1 / in2 are bf16 tensors (dy fp32)
// ... and with the operands' roles exchanged for <= 2 input channels (first layers: seg 1 -> 8, reg 1 + 1 -> 16)
bool da_conv3_fewcin_wgrad_supported(int C1, int C2, int Cout, int stride);
int da_conv3_fewcin_wgrad(const float* in1, int C1, const float* in2, int C2, const float* dy, float* dw_tio,
                          int N, int D, int H, int W, int Cout, void* ws, size_t ws_bytes, hipStream_t st, int dy_bf16 = 0);     // dy_bf16: dy is a bf16 tensor (in1 / in2 fp32)

// ---- kept packed operands of the kernel families OUTSIDE the split matrix kernels (folded up-sampling, native stride-2, flow, thin) -----------------
// Each of them packs its weights into the call's workspace before its main kernel: a 5 - 20 us launch in the dependent chain of a layer, every call, for
// data that only changes when the optimiser steps (a registration step: 22 of them, 0.2 ms of 4.3).  da_conv3d_k3_prepack_any (conv3d.hip) fills a
// caller-owned buffer, da_conv3d_k3_use_prepacked_any hands it to the NEXT call on the same weights.  A family calls da_pp_lookup where it would pack:
//   .buf != nullptr: the packed operand lives there -- pack into it first when .fill, it is ready otherwise;  .buf == nullptr: pack into the workspace as always;
//   .only: return right after the pack stage (a fill call: dummy tensor pointers, nothing else may be launched).
// tag: the family's own id for (kernel, direction), so that a hand-over meant for the forward is never taken by the data gradient of the same weights.
struct DaKeptPack { unsigned char* buf; int fill; int only; };
DaKeptPack da_pp_lookup(const float* w_tio, size_t need, int tag);
void da_pp_drop_handover();      // at the end of every public entry that may have been handed a pack: an unconsumed hand-over serves no later call
struct DaPpScope { ~DaPpScope() { da_pp_drop_handover(); } };
enum { DA_PP_UP_FWD = 1, DA_PP_UP_DGRAD, DA_PP_S2N_FWD, DA_PP_S2N_DGRAD, DA_PP_FM_FWD, DA_PP_FM_DGRAD, DA_PP_THIN, DA_PP_THIN_FLIP };
bool da_conv3_s2_is_native(int Cin, int Cout, int N, int D, int H, int W);      // conv3d_s2.hip: the stride-2 layer runs conv3d_s2n.hip
