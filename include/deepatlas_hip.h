/*
 * deepatlas_hip.h -- C ABI of libdeepatlas_hip.so, the MI355X (gfx950) native hot path for
 * uncbiag/DeepAtlas' 3D volumetric training step.
 *
 * The reference has no FFI layer: its hot path is Python (L2) calling PyTorch ATen ops (L1)
 * (SURVEY.md §1, §8b).  This header is the seam a maintainer would bind instead of those ATen
 * calls; each entry cites the reference call site(s) it replaces (file:line in uncbiag/DeepAtlas).
 *
 * Conventions
 *   - every activation tensor is fp32, channels-last 3D ("NDHWC"): [N][D][H][W][C], dense.
 *     (torch side: a N x C x D x H x W tensor with torch.channels_last_3d strides.)
 *   - raw device pointers, explicit dims, no hidden allocation: scratch memory is passed in as
 *     (ws, ws_bytes); da_*_ws_bytes() returns the size a call needs.
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls only enqueue work.
 *   - return value: 0 on success, otherwise a hipError_t, or DA_ERR_* (negative) for bad arguments.
 *   - 3x3x3 conv weights use the "TIO" layout [27 taps (kd,kh,kw)][Cin][Cout]; 2x2x2 transposed-conv
 *     weights [8 taps][Cin][Cout]; 1x1x1 weights [Cin][Cout].  da_w_* convert from/to the PyTorch
 *     state_dict layouts ([Cout][Cin][k][k][k], ConvTranspose [Cin][Cout][k][k][k]).
 */
#ifndef DEEPATLAS_HIP_H
#define DEEPATLAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DA_ERR_BADARG   (-1)
#define DA_ERR_WS_SMALL (-2)
#define DA_ERR_UNSUPPORTED (-3)

/* library / device info */
int  da_version(void);
int  da_device_info(int* cu_count, int* wave_size, size_t* hbm_bytes, char* arch, int arch_len);

/* ---- weight layout transforms (tiny) -------------------------------------------------------- */
/* [Cout][Cin][k^3] (nn.Conv3d, unets.py:30,36,250; modules.py:48; voxel_morph.py:57) <-> [k^3][Cin][Cout] */
int da_w_oik_to_tio(const float* w_oik, float* w_tio, int Cout, int Cin, int K3, void* stream);
int da_w_tio_to_oik(const float* w_tio, float* w_oik, int Cout, int Cin, int K3, void* stream);
/* [Cin][Cout][k^3] (nn.ConvTranspose3d, unets.py:49,55) <-> [k^3][Cin][Cout] */
int da_w_iok_to_tio(const float* w_iok, float* w_tio, int Cin, int Cout, int K3, void* stream);
int da_w_tio_to_iok(const float* w_tio, float* w_iok, int Cin, int Cout, int K3, void* stream);
/* ConvTranspose3d(k=3, s=1, p=1) weights [Cin][Cout][27] (unets.py:88-96 `UNet.decoder`): the same op as a 3x3x3 convolution with
 * flipped taps, so it runs on the da_conv3d_k3_* entries after this re-layout. */
int da_w_iok_flip_to_tio(const float* w_iok, float* w_tio, int Cin, int Cout, int K3, void* stream);
int da_w_tio_to_iok_flip(const float* w_tio, float* w_iok, int Cin, int Cout, int K3, void* stream);
/* the three gradient conversions, accumulating: dst += layout(w_tio) (dst: the parameter's .grad inside the optimiser's flat bucket) */
int da_w_tio_to_oik_acc(const float* w_tio, float* w_oik, int Cout, int Cin, int K3, void* stream);
int da_w_tio_to_iok_acc(const float* w_tio, float* w_iok, int Cin, int Cout, int K3, void* stream);
int da_w_tio_to_iok_flip_acc(const float* w_tio, float* w_iok, int Cin, int Cout, int K3, void* stream);

/* ---- 3x3x3 convolution, padding 1, stride 1|2 (rows a1, a7, a9) ----------------------------- */
/* replaces nn.Conv3d(k=3,p=1) forward: unets.py:30,36; modules.py:48,56; voxel_morph.py:57,82.
 * Input = channel-concat(in1[C1], in2[C2]) without materialising it (torch.cat at unets.py:275,
 * voxel_morph.py:65,74,76,78,82); in2 may be NULL with C2 = 0.
 * out[N][Do][Ho][Wo][Cout], Do = (D-1)/stride+1.  bias may be NULL.
 * act_slope < 0: no activation; == 0: ReLU (modules.py:58); > 0: LeakyReLU(slope) fused in the epilogue.
 * stats (optional, may be NULL): per-channel double [2][Cout] (sum, sum of squares) of the PRE-activation
 * output, ACCUMULATED into by the kernel's epilogue is not done here; see da_bn_stats. */
size_t da_conv3d_k3_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int stride);
int da_conv3d_k3_fwd(const float* in1, int C1, const float* in2, int C2,
                     const float* w_tio, const float* bias, float* out,
                     int N, int D, int H, int W, int Cout, int stride, float act_slope,
                     void* ws, size_t ws_bytes, void* stream);
/* Forward of a conv that feeds a train-mode BatchNorm: same as da_conv3d_k3_fwd (no activation) and, when the MFMA path
 * serves the layer, the epilogue also writes per-workgroup partial sums stats_partial[nparts][2][Cout] (double: sum, sum of
 * squares of the output) so the BN statistics need no extra pass over the tensor.  *stats_nparts = 0 means "not fused":
 * run da_bn_train_stats.  stats_capacity = number of [2][Cout] slots available (>= 512). */
int da_conv3d_k3_fwd_bnstats(const float* in1, int C1, const float* in2, int C2,
                             const float* w_tio, const float* bias, float* out,
                             int N, int D, int H, int W, int Cout, int stride,
                             double* stats_partial, int stats_capacity, int* stats_nparts,
                             void* ws, size_t ws_bytes, void* stream);
/* data gradient (autograd of the above): dx = concat(dx1[C1], dx2[C2]); D,H,W are the INPUT dims. */
int da_conv3d_k3_dgrad(const float* dy, const float* w_tio, float* dx1, int C1, float* dx2, int C2,
                       int N, int D, int H, int W, int Cout, int stride,
                       void* ws, size_t ws_bytes, void* stream);
/* weight / bias gradient: dw_tio[27][C1+C2][Cout], dbias[Cout] (may be NULL). */
int da_conv3d_k3_wgrad(const float* in1, int C1, const float* in2, int C2, const float* dy,
                       float* dw_tio, float* dbias,
                       int N, int D, int H, int W, int Cout, int stride,
                       void* ws, size_t ws_bytes, void* stream);

/* Forward / weight gradient of the same convolution with an INPUT PROLOGUE (the reference's Conv3d -> BatchNorm3d -> LeakyReLU
 * -> Conv3d chains, unets.py:24-39, 259-278): in1 / in2 may be the RAW output of the producing convolution; its BatchNorm
 * scale[C] / shift[C] (da_bn_train_stats* / da_bn_eval_affine) and activation slope are applied while the tile is staged, with
 * the arithmetic of da_bn_act_fwd (bit-identical to materialising the activated tensor).  proN_scale == NULL: input N is an
 * ordinary tensor.  stats_* as in da_conv3d_k3_fwd_bnstats (NULL / capacity < 512: no statistics, act_slope is applied).
 * Stride 1 on the matrix cores only: DA_ERR_UNSUPPORTED tells the caller to apply da_bn_act_fwd and use the plain entries. */
int da_conv3d_k3_fwd_pro(const float* in1, int C1, const float* pro1_scale, const float* pro1_shift, float pro1_slope,
                         const float* in2, int C2, const float* pro2_scale, const float* pro2_shift, float pro2_slope,
                         const float* w_tio, const float* bias, float* out,
                         int N, int D, int H, int W, int Cout, float act_slope,
                         double* stats_partial, int stats_capacity, int* stats_nparts,
                         void* ws, size_t ws_bytes, void* stream);
int da_conv3d_k3_wgrad_pro(const float* in1, int C1, const float* pro1_scale, const float* pro1_shift, float pro1_slope,
                           const float* in2, int C2, const float* pro2_scale, const float* pro2_shift, float pro2_slope,
                           const float* dy, float* dw_tio,
                           int N, int D, int H, int W, int Cout,
                           void* ws, size_t ws_bytes, void* stream);

/* ---- nearest x2 up-sampling folded into the 3x3x3 convolution that consumes it (row a8 + a7; voxel_morph.py:72-80: `F.interpolate(x,
 * size=skip.shape)` -- default mode 'nearest' -- then modules.convBlock, modules.py:48,56-58).  s1 / s2: the one or two COARSE source
 * tensors [N][Dc][Hc][Wc][C1|C2] (s2 / C2 = NULL / 0 for a single input; the reference concatenates, voxel_morph.py:73,75); out and dy
 * live on the fine grid [N][2Dc][2Hc][2Wc][Cout]; w_tio / dw_tio [27][C1+C2][Cout] as for da_conv3d_k3_*.  Exact x2 only (callers use
 * da_upsample_nearest_* + da_conv3d_k3_* otherwise), split matrix mode only (da_upconv3d_k3_supported returns 0 otherwise; the entry
 * points then return DA_ERR_UNSUPPORTED).  Per output parity the layer is a 2x2x2-tap convolution on the coarse grid with summed weights
 * (conv3d_up2.hip): neither the up-sampled tensor nor its gradient is materialised. */
int da_upconv3d_k3_supported(int C1, int C2, int Cout);
size_t da_upconv3d_k3_ws_bytes(int N, int Dc, int Hc, int Wc, int Cin, int Cout);
int da_upconv3d_k3_fwd(const float* s1, int C1, const float* s2, int C2, const float* w_tio, const float* bias, float* out,
                       int N, int Dc, int Hc, int Wc, int Cout, float act_slope, void* ws, size_t ws_bytes, void* stream);
int da_upconv3d_k3_dgrad(const float* dy, const float* w_tio, float* dx1, int C1, float* dx2, int C2,
                         int N, int Dc, int Hc, int Wc, int Cout, void* ws, size_t ws_bytes, void* stream);
int da_upconv3d_k3_wgrad(const float* s1, int C1, const float* s2, int C2, const float* dy, float* dw_tio,
                         int N, int Dc, int Hc, int Wc, int Cout, void* ws, size_t ws_bytes, void* stream);

/* test/diagnostic knob: force the direct (VALU) kernels instead of the MFMA implicit-GEMM path (also env
 * DA_CONV_DIRECT=1); returns the previous setting.  Used by the GPU tests to A/B the two implementations. */
int da_set_conv_direct(int on);

/* bf16 matrix mode (BASELINE config 5; the reference itself is fp32-only, train_seg.py has no autocast): when on, the 3x3x3
 * convolutions (forward, data gradient, weight gradient) round their operands to bf16 while staging them and run on
 * v_mfma_f32_16x16x16_bf16 with fp32 accumulation; tensors in HBM, BatchNorm, losses and the optimiser stay fp32.
 * Off by default (exact fp32 v_mfma_f32_16x16x4_f32); returns the previous setting. */
int da_set_matrix_bf16(int on);
/* Matrix mode of the 3x3x3 convolutions (forward, data gradient, weight gradient); returns the previous mode, -1 for an unknown one.
 *   0  fp32 operands on v_mfma_f32_16x16x4_f32 -- one fmaf per product, the arithmetic of the reference's nn.Conv3d on the CPU
 *      (lib/network_factory/modules.py:48);
 *   1  = da_set_matrix_bf16(1): operands ROUNDED to bf16 (not fp32-accurate);
 *   2  "split": every staged tile is scaled by a power of two (its largest magnitude into [2^14, 2^15)) and every fp32 operand split into
 *      two fp16 terms (x s = h + l to 2^-22 |x s|); a product is formed from three of the four partial products (h l', l h', h h') on
 *      v_mfma_f32_16x16x32_f16 with fp32 accumulation, the accumulators rescaled by exact powers of two between tiles.  Per product the
 *      error bound is 2^-21 + 2^-22 |x y| (typically one fp32 rounding); over the sums of a convolution the result is closer to double
 *      than mode 0's fmaf chain (tests/test_gpu_split.py), at 3/16 of the matrix-pipe time.  Tensors in HBM stay fp32
 *      (deepatlas_amd/csrc/split_f16.h). */
int da_set_matrix_mode(int mode);

/* ---- 1x1x1 convolution (segmentation head, row a5; unets.py:249-250) ------------------------- */
/* scratch for the packed weights of the 1x1 / transposed-conv forward and data-gradient launchers */
size_t da_pointwise_ws_bytes(int ntaps, int Cin, int Cout);
int da_conv1x1_fwd(const float* in, const float* w_io, const float* bias, float* out,
                   long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* Head with an INPUT PROLOGUE (see da_conv3d_k3_fwd_pro): `in` is the raw output of the last decoder convolution and
 * act(in * pro_scale + pro_shift) is what the 1x1x1 convolution consumes; same arithmetic as da_bn_act_fwd followed by the plain
 * entries.  Channel counts in multiples of 16 (matrix-core kernels), else DA_ERR_UNSUPPORTED. */
int da_conv1x1_fwd_pro(const float* in, const float* pro_scale, const float* pro_shift, float pro_slope,
                       const float* w_io, const float* bias, float* out,
                       long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int da_conv1x1_wgrad_pro(const float* in, const float* pro_scale, const float* pro_shift, float pro_slope,
                         const float* dy, float* dw_io, float* dbias,
                         long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int da_conv1x1_dgrad(const float* dy, const float* w_io, float* dx, long long M, int Cin, int Cout,
                     void* ws, size_t ws_bytes, void* stream);
size_t da_conv1x1_wgrad_ws_bytes(long long M, int Cin, int Cout);
int da_conv1x1_wgrad(const float* in, const float* dy, float* dw_io, float* dbias,
                     long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

/* ---- 2x2x2 stride-2 transposed convolution (row a3; unets.py:49,55,240-241) ------------------ */
/* out[N][2D][2H][2W][Cout] = bias + sum_ci in[N][D][H][W][ci] * w_tio[tap(i,j,k)][ci][co] */
int da_deconv_k2s2_fwd(const float* in, const float* w_tio, const float* bias, float* out,
                       int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* Same, with the BatchNorm3d partial sums of the output accumulated in the epilogue (unets.py:49-51: ConvTranspose3d ->
 * BatchNorm3d): stats_partial[stats_nparts][2][Cout] doubles for da_bn_train_stats_from_partials; needs
 * stats_capacity >= ceil(N*D*H*W / 256).  *stats_nparts == 0 on return: no statistics were produced (shape / capacity), run
 * da_bn_train_stats over the output instead. */
int da_deconv_k2s2_fwd_bnstats(const float* in, const float* w_tio, const float* bias, float* out,
                               int N, int D, int H, int W, int Cin, int Cout,
                               double* stats_partial, int stats_capacity, int* stats_nparts,
                               void* ws, size_t ws_bytes, void* stream);
int da_deconv_k2s2_dgrad(const float* dy, const float* w_tio, float* dx,
                         int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* Fused backward of the up-sampler block ConvTranspose3d(k2, s2) -> BatchNorm3d(train) -> LeakyReLU/ReLU (autograd of unets.py:49-52).  gout = gradient
 * with respect to the ACTIVATED output, y = the raw transposed-conv output, both [N][2D][2H][2W][Cout]; mean / rstd / scale / shift = the block's statistics
 * rows; in = the block's input [N][D][H][W][Cin].  One pass over (gout, y) after the sums (one reduction pass, or `pre`[pre_n][2][Cout] = sums accumulated by
 * the producer of gout as for da_bn_act_bwd_dbias_pre): dx [N][D][H][W][Cin], dw_tio [8][Cin][Cout], dbias[Cout] (may be NULL), dgamma[Cout], dbeta[Cout].
 * The tensor dy = d loss / d y is never written.  DA_ERR_UNSUPPORTED unless Cin = Cout = 32 (callers fall back to da_bn_act_bwd_dbias +
 * da_deconv_k2s2_dgrad + da_deconv_k2s2_wgrad). */
size_t da_deconv_k2s2_bn_bwd_ws_bytes(int N, int D, int H, int W, int Cin, int Cout);
int da_deconv_k2s2_bn_bwd(const float* gout, const float* y, const float* mean, const float* rstd, const float* scale, const float* shift, float act_slope,
                          const float* in, const float* w_tio, float* dx, float* dw_tio, float* dbias, float* dgamma, float* dbeta,
                          int N, int D, int H, int W, int Cin, int Cout, const double* pre, int pre_n,
                          void* ws, size_t ws_bytes, void* stream);
size_t da_deconv_k2s2_wgrad_ws_bytes(int N, int D, int H, int W, int Cin, int Cout);
int da_deconv_k2s2_wgrad(const float* in, const float* dy, float* dw_tio, float* dbias,
                         int N, int D, int H, int W, int Cin, int Cout,
                         void* ws, size_t ws_bytes, void* stream);

/* ---- BatchNorm3d + LeakyReLU (rows a1, a3; unets.py:31,51 + :5-6) ---------------------------- */
/* Train-mode statistics over M = N*D*H*W rows of x[M][C]: writes mean[C], rstd[C] (1/sqrt(biased var+eps)),
 * scale[C] = gamma*rstd, shift[C] = beta - mean*scale, and updates running_mean/var in place with
 * `momentum` and the UNBIASED variance (nn.BatchNorm3d semantics).  running_* may be NULL. */
size_t da_bn_ws_bytes(long long M, int C);
int da_bn_train_stats(const float* x, long long M, int C, const float* gamma, const float* beta,
                      float eps, float momentum, float* running_mean, float* running_var,
                      float* mean, float* rstd, float* scale, float* shift,
                      void* ws, size_t ws_bytes, void* stream);
/* Same result from partial sums produced by da_conv3d_k3_fwd_bnstats. */
int da_bn_train_stats_from_partials(const double* partial, int nparts, long long M, int C,
                                    const float* gamma, const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var,
                                    float* mean, float* rstd, float* scale, float* shift, void* stream);
/* Eval-mode affine from running statistics. */
int da_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, int C, float* mean, float* rstd, float* scale, float* shift, void* stream);
/* y = act(x*scale + shift); act_slope as in da_conv3d_k3_fwd. */
int da_bn_act_fwd(const float* x, const float* scale, const float* shift, float act_slope, float* y,
                  long long M, int C, void* stream);
/* Backward of act(BN_train(x)): given dy (grad wrt y) and the saved x, mean, rstd, gamma, scale, shift:
 * dx[M][C], dgamma[C], dbeta[C].  train != 0: full batch-statistics backward; train == 0: dx = dz*scale. */
int da_bn_act_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                  const float* scale, const float* shift, float act_slope, int train,
                  float* dx, float* dgamma, float* dbeta, long long M, int C,
                  void* ws, size_t ws_bytes, void* stream);
/* Same, additionally writing dxsum[C] = per-channel column sums of dx: the bias gradient of the convolution (or transposed
 * convolution) that produced x, fused into the apply pass so the conv weight-gradient call can skip its own pass over dy. */
int da_bn_act_bwd_dbias(const float* dy, const float* x, const float* mean, const float* rstd,
                        const float* scale, const float* shift, float act_slope, int train,
                        float* dx, float* dgamma, float* dbeta, float* dxsum, long long M, int C,
                        void* ws, size_t ws_bytes, void* stream);
/* da_bn_act_bwd_dbias without its reduction pass over (dy, x): the two sums come from the kernel that PRODUCED dy, which accumulated them in its
 * epilogue -- pre[pre_n][2][C] doubles = (sum dz, sum dz (x - mean)) per workgroup, dz = dy act'(x scale + shift) (da_head_dice_bwd_bst).
 * train only.  autograd of nn.BatchNorm3d + nn.LeakyReLU (unets.py:31-32). */
int da_bn_act_bwd_dbias_pre(const float* dy, const float* x, const float* mean, const float* rstd,
                            const float* scale, const float* shift, float act_slope, int train,
                            float* dx, float* dgamma, float* dbeta, float* dxsum, long long M, int C,
                            const double* pre, int pre_n, void* ws, size_t ws_bytes, void* stream);
/* Backward of a bare activation from its OUTPUT y (ReLU / LeakyReLU): dx = dy * (y > 0 ? 1 : slope). */
int da_act_bwd(const float* dy, const float* y, float act_slope, float* dx, long long numel, void* stream);
/* Per-channel column sum of x[M][C] -> out[C] (bias gradients). */
/* backward of a conv block without BatchNorm (modules.py:56-58 conv -> act): dx = (g1 [+ g2]) * act'(y) and dbias = column sums of dx
 * in ONE pass.  g2 (may be NULL): second incoming gradient of a block whose output has two consumers (skip connection); y: the
 * block's ACTIVATED output (unused when act_slope < 0); dbias may be NULL; ws as da_bn_ws_bytes(M, C). */
int da_act_bwd_add_dbias(const float* g1, const float* g2, const float* y, float act_slope, float* dx, float* dbias,
                         long long M, int C, void* ws, size_t ws_bytes, void* stream);
int da_colsum(const float* x, long long M, int C, float* out, void* ws, size_t ws_bytes, void* stream);
/* da_act_bwd_add_dbias in two halves (same call site, modules.py:56-58): the big pass leaves its per-block double partial sums in the
 * caller-owned `partial` (da_bn_ws_bytes(M, C) bytes; *nparts = number of partial sets written), and da_colsum_finish reduces them to
 * out[C] (accumulate != 0: out[c] += sum) -- possibly on another stream, once the first call has completed there. */
int da_act_bwd_add_partial(const float* g1, const float* g2, const float* y, float act_slope, float* dx,
                           long long M, int C, void* partial, size_t partial_bytes, int* nparts, void* stream);
int da_colsum_finish(const void* partial, int nparts, int C, float* out, int accumulate, void* stream);

/* ---- MaxPool3d(2) (row a2; unets.py:230,267) ------------------------------------------------- */
int da_maxpool2_fwd(const float* x, float* y, int N, int D, int H, int W, int C, void* stream);
/* MaxPool3d(2) of a tensor whose BatchNorm + activation is still pending (see da_conv3d_k3_fwd_pro): one pass writes the activated
 * tensor `act` (the skip connection of unets.py:266-267) and its pooled version `y`; arithmetic of da_bn_act_fwd + da_maxpool2_fwd.
 * Even D, H, W and C % 4 == 0, else DA_ERR_UNSUPPORTED. */
int da_maxpool2_fwd_pro(const float* x, const float* pro_scale, const float* pro_shift, float pro_slope, float* act, float* y,
                        int N, int D, int H, int W, int C, void* stream);
/* dx (input-sized) from dy and the saved input x; gradient goes to the first maximum in (d,h,w) scan order. */
int da_maxpool2_bwd(const float* dy, const float* x, float* dx, int N, int D, int H, int W, int C, void* stream);
/* dx = gskip + maxpool_bwd(dy): the pooled tensor also feeds a skip connection (unets.py:266-267,275) */
int da_maxpool2_bwd_add(const float* dy, const float* x, const float* gskip, float* dx, int N, int D, int H, int W, int C, void* stream);
/* The same (gskip may be NULL) on the RAW tensor da_maxpool2_fwd_pro pooled, whose BatchNorm + activation was applied on the fly: x_raw = the producer's raw
 * conv output, stats4 = its statistics rows [mean | rstd | scale | shift][C], slope its activation.  The arg-max is that of the activated values (formed again
 * with the forward's expression); besides dx the kernel accumulates the PRODUCER's BatchNorm-backward sums, bst[*bst_n][2][C] doubles =
 * (sum dz, sum dz (x - mean)), dz = dx act'(x scale + shift), for da_bn_act_bwd_dbias_pre.  Even D, H, W, C / 4 a power of two, bst_cap >= 1024; else
 * DA_ERR_UNSUPPORTED.  autograd of nn.MaxPool3d(2) + the skip connection behind conv -> BatchNorm -> LeakyReLU (unets.py:30-37, 266-267). */
int da_maxpool2_bwd_bst(const float* dy, const float* x_raw, const float* gskip, float* dx, int N, int D, int H, int W, int C,
                        const float* stats4, float slope, double* bst, int bst_cap, int* bst_n, void* stream);

/* ---- nearest-neighbour up-sampling to a given size (row a8; voxel_morph.py:72,74,76,80) ------ */
int da_upsample_nearest_fwd(const float* x, float* y, int N, int D, int H, int W, int C,
                            int Do, int Ho, int Wo, void* stream);
int da_upsample_nearest_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C,
                            int Do, int Ho, int Wo, void* stream);

/* ---- deformation-field trilinear warp (rows a9-a10; voxel_morph.py:85-91, lib/utils.py:89-102) */
/* deform = disp + identity (identity generated in-kernel, channel order (x,y,z) = (W,H,D) axis, normalised
 * to [-1,1]); out = grid_sample(src, deform, bilinear, zeros, align_corners=True).
 * src[N][D][H][W][C], disp[N][D][H][W][3], deform (may be NULL) [N][D][H][W][3], out[N][D][H][W][C]. */
int da_warp_fwd(const float* src, const float* disp, float* deform, float* out,
                int N, int D, int H, int W, int C, void* stream);
/* gradients: d_disp[N][D][H][W][3] (may be NULL) and d_src (may be NULL; must be ZERO-FILLED by the caller,
 * accumulated with float atomics). */
int da_warp_bwd(const float* dout, const float* src, const float* disp, float* d_disp, float* d_src,
                int N, int D, int H, int W, int C, void* stream);
/* lib/utils.py:78-102 get_identity_transform(_batch): out[3][D][H][W] (reference layout, channel-first) */
/* warp of one-hot(labels) without materialising the one-hot tensor (joint step): out [N][D][H][W][C];
 * labels uint8 (1) / int64 (8) [N][D][H][W]; backward w.r.t. the displacement only (labels carry no gradient). */
int da_warp_labels_fwd(const void* labels, int label_bytes, const float* disp, float* out,
                       int N, int D, int H, int W, int C, void* stream);
int da_warp_labels_bwd(const float* dout, const void* labels, int label_bytes, const float* disp, float* d_disp,
                       int N, int D, int H, int W, int C, void* stream);
int da_identity_grid(float* out, int D, int H, int W, int normalize, void* stream);
/* ---- fused anatomy losses of the joint step (parts: lib/loss.py:410-476 Dice, voxel_morph.py:90-91 warp) ------
 * Dice's input gradient is g[v][c] = coef[0][n][c] [St[v] == c] + coef[1][n][c] (coef as written by da_dice_fwd), which lets both
 * anatomy terms skip their 32-channel intermediate tensors:
 *  registration phase: loss = Dice(warp(onehot(lab_m), id + disp), onehot(lab_t)) straight from the two label maps (no warped
 *    one-hot, no gradient tensor); bwd writes d loss / d disp.
 *  segmentation phase: the adjoint warp of g is coef[1][c] A[u] + coef[0][c] B[u][c] with A = W^T 1 ([N][V]) and
 *    B = W^T onehot(lab_t), stored CLASS-MAJOR ([N][C][V]: the atomics of a wave then fall into consecutive floats of one plane) --
 *    da_warp_adjoint_labels zero-fills and scatters B (8 float atomics per voxel instead of 8 C);
 *    A[u] = sum_c B[u][c] is formed on the fly, the optional array A only collects the weights of voxels whose target label is outside
 *    [0, C) (pass NULL when the labels are known to be valid) -- and da_seg_anat_dlogits forms dlogits ([N][V][C]) = d(loss_sup + loss_anat) / d logits from prob and B through the softmax Jacobian
 *    (prob = softmax(logits); coef_sup / lab_m / dloss_sup NULL when there is no supervised Dice term). */
size_t da_label_warp_dice_ws_bytes(int N, int C);
int da_label_warp_dice_fwd(const void* lab_m, int lab_m_bytes, const void* lab_t, int lab_t_bytes, const float* disp,
                           int N, int D, int H, int W, int C, int weight_type, int no_bg, float eps,
                           float* loss, float* coef /*[2][N][C]*/, void* ws, size_t ws_bytes, void* stream);
int da_label_warp_dice_bwd(const void* lab_m, int lab_m_bytes, const void* lab_t, int lab_t_bytes, const float* disp,
                           const float* coef, const float* dloss, float* d_disp, int N, int D, int H, int W, int C, void* stream);
/* segmentation phase, forward: (a) Dice(softmax(src), labels) as da_dice_fwd(softmax = 1) AND prob = softmax(src) written in the same pass
 * (the warp below needs the probabilities: one pass over the logits instead of da_dice_fwd + da_softmax_fwd); (b) Dice(warp(prob, id + disp),
 * onehot(lab_t)) without writing the warped tensor (da_warp_fwd + da_dice_fwd minus one write and one read of N V C floats).  Both return
 * DA_ERR_UNSUPPORTED for class counts whose 4-channel groups are not a power of two; callers then run the separate entries. */
int da_softmax_dice_fwd(const float* src, const void* labels, int label_bytes, float* prob,
                        int N, long long V, int C, int weight_type, int no_bg, float eps,
                        float* loss, float* coef /*[2][N][C]*/, void* ws /* da_dice_ws_bytes */, size_t ws_bytes, void* stream);
size_t da_warp_dice_ws_bytes(int N, int C);
int da_warp_dice_fwd(const float* src, const float* disp, const void* lab_t, int lab_t_bytes,
                     int N, int D, int H, int W, int C, int weight_type, int no_bg, float eps,
                     float* loss, float* coef /*[2][N][C]*/, void* ws, size_t ws_bytes, void* stream);
int da_warp_adjoint_labels(const void* lab_t, int lab_t_bytes, const float* disp, float* A, float* B,
                           int N, int D, int H, int W, int C, void* stream);
int da_seg_anat_dlogits(const float* prob, const void* lab_m, int lab_m_bytes, const float* A, const float* B, float* dlogits,
                        const float* coef_sup, const float* coef_anat, const float* dloss_sup, const float* dloss_anat,
                        int N, long long V, int C, void* stream);
/* deterministic d_src (parity runs): the same scatter as da_warp_bwd's d_src, accumulated in 64-bit fixed point with integer atomics
 * (order-independent, hence run-to-run bit-identical), scale = power of two from max|dout|.  d_src is OVERWRITTEN (no pre-zeroing). */
size_t da_warp_bwd_dsrc_det_ws_bytes(int N, int D, int H, int W, int C);
int da_warp_bwd_dsrc_det(const float* dout, const float* disp, float* d_src, int N, int D, int H, int W, int C,
                         void* ws, size_t ws_bytes, void* stream);

/* ---- fused softmax + Dice loss (row a11; lib/loss.py:410-476, lib/transforms.py:675-689) ------ */
/* src[N][V][C] logits (softmax != 0) or probabilities; target: labels (label_bytes = 1 uint8 | 8 int64,
 * [N][V]) or, when soft_target != NULL, a soft target [N][V][C] (loss.py:435-436).
 * weight_type: 0 Uniform, 1 Simple, 2 Volume.  Writes loss[1] and coef[2][N][C] for the backward. */
size_t da_dice_ws_bytes(int N, long long V, int C);
int da_dice_fwd(const float* src, const void* labels, int label_bytes, const float* soft_target,
                int N, long long V, int C, int softmax, int weight_type, int no_bg, float eps,
                float* loss, float* coef, void* ws, size_t ws_bytes, void* stream);
/* d_src[N][V][C] = dloss * dL/dsrc; d_soft (may be NULL) = dloss * dL/dsoft_target is not provided. */
int da_dice_bwd(const float* src, const void* labels, int label_bytes, const float* soft_target,
                const float* coef, const float* dloss, float* d_src,
                int N, long long V, int C, int softmax, void* stream);
/* softmax over the channel axis of x[M][C] (joint step: probabilities to warp), and its backward. */
int da_softmax_fwd(const float* x, float* y, long long M, int C, void* stream);
int da_softmax_bwd(const float* dy, const float* y, float* dx, long long M, int C, void* stream);
/* lib/transforms.py:675-689 mask_to_one_hot: labels[N][V] -> out[N][V][C] float */
int da_one_hot(const void* labels, int label_bytes, float* out, long long M, int C, void* stream);

/* ---- fused segmentation head + softmax + Dice (rows a5 + a11: unets.py:249-250 followed by lib/loss.py:410-476 with softmax=True and
 *      an index target).  loss = Dice(softmax(x W + b), labels) WITHOUT writing the logits: the 16 -> 32 head is recomputed in the
 *      backward pass, which turns Dice's gradient straight into dx, dW and db.  x [N][V][Cin] channels-last; w_io [Cin][C];
 *      pro_scale / pro_shift (may be NULL) + pro_slope: BatchNorm + activation of the producer still to be applied to x (as in
 *      da_conv1x1_fwd_pro); coef [2][N][C] as da_dice_fwd.  Shapes: Cin in {16, 64}, C in {16, 32}; others -> DA_ERR_UNSUPPORTED
 *      (callers then run da_conv1x1_* + da_dice_*).  dx is the gradient with respect to the ACTIVATED input. */
size_t da_head_dice_ws_bytes(int N, long long V, int Cin, int C);
int da_head_dice_fwd(const float* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                     const float* w_io, const float* bias, const void* labels, int label_bytes,
                     int N, long long V, int Cin, int C, int weight_type, int no_bg, float eps,
                     float* loss, float* coef, void* ws, size_t ws_bytes, void* stream);
int da_head_dice_bwd(const float* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                     const float* w_io, const float* bias, const void* labels, int label_bytes,
                     const float* coef, const float* dloss, float* dx, float* dw_io, float* dbias,
                     int N, long long V, int Cin, int C, void* ws, size_t ws_bytes, void* stream);
/* da_head_dice_bwd that also accumulates the BatchNorm-backward sums of the layer that produced x (x = its raw conv output; pro_scale / pro_shift /
 * pro_mean = rows of its batch statistics): bst[*bst_n][2][Cin] doubles for da_bn_act_bwd_dbias_pre.  *bst_n = 0: shape without that epilogue
 * (Cin != 16, bst_cap < 1024) -- run da_bn_act_bwd_dbias as usual. */
int da_head_dice_bwd_bst(const float* x, const float* pro_scale, const float* pro_shift, float pro_slope, const float* pro_mean,
                         const float* w_io, const float* bias, const void* labels, int label_bytes,
                         const float* coef, const float* dloss, float* dx, float* dw_io, float* dbias,
                         int N, long long V, int Cin, int C, double* bst, int bst_cap, int* bst_n, void* ws, size_t ws_bytes, void* stream);

/* da_conv3d_k3_dgrad of a layer whose FIRST input was act(BN(y)) applied on the fly: y = the producer's raw conv output, stats4 = its statistics rows
 * [mean | rstd | scale | shift][C1], slope its activation.  dx1 / dx2 as always (gradients with respect to the ACTIVATED tensors); the epilogue also accumulates
 * the PRODUCER's BatchNorm-backward sums, bst[*bst_n][2][C1] doubles = (sum dz, sum dz (y - mean)), dz = dx1 act'(y scale + shift), for
 * da_bn_act_bwd_dbias_pre -- no reduction pass over (dx1, y).  Shapes: one input of <= 16 channels (dx2 NULL, C2 0), or the decoder's concat layer (C1 = 32
 * up-sampled channels, C2 = 16 skip channels; unets.py:275).  Split matrix mode only; DA_ERR_UNSUPPORTED: run da_conv3d_k3_dgrad and the usual backward.
 * autograd of nn.Conv3d / nn.ConvTranspose3d -> nn.BatchNorm3d -> nn.LeakyReLU chains (unets.py:30-37, 50-53). */
int da_conv3d_k3_dgrad_bst(const float* dy, const float* w_tio, float* dx1, int C1, float* dx2, int C2, int N, int D, int H, int W, int Cout,
                           const float* y, const float* stats4, float slope, double* bst, int bst_cap, int* bst_n,
                           void* ws, size_t ws_bytes, void* stream);

/* ---- packed convolution operands kept across calls (split matrix mode; conv3d_mfma.hip) ----------------------------------------------
 * A 3x3x3 convolution call first packs its weights for the matrix cores and writes its tile table (a 10-us launch in front of every matrix kernel).  Both
 * only change when the weights do (torch.optim.Adam.step in the reference's loop, models/segmentation.py:157): the caller may keep them.
 *   da_conv3d_k3_pack_bytes      bytes of one region for a C1 + C2 = Cin -> Cout layer on an N x D x H x W grid (0: no such kernel for the shape)
 *   da_conv3d_k3_prepack         fill b0 (and b1: the data gradient of a 32 + 16 concat layer runs two launches) for the forward (dgrad = 0) or the data
 *                                gradient (dgrad = 1); *used = regions filled, 0 = this shape / matrix mode keeps nothing.  Only enqueues on `stream`.
 *   da_conv3d_k3_use_prepacked   the NEXT da_conv3d_k3_{fwd,fwd_bnstats,fwd_pro,dgrad} call of this thread on `w_tio` skips its pack launch and reads b0 / b1
 *                                (one call; dropped when the next call is on other weights).  The caller orders the fill before the use (stream / event). */
size_t da_conv3d_k3_pack_bytes(int N, int D, int H, int W, int Cin, int Cout);
int da_conv3d_k3_prepack(const float* w_tio, int C1, int C2, int Cout, int dgrad, int N, int D, int H, int W,
                         void* b0, size_t n0, void* b1, size_t n1, int* used, void* stream);
void da_conv3d_k3_use_prepacked(const float* w_tio, const void* b0, size_t n0, const void* b1, size_t n1);
/* da_conv3d_k3_prepack for n layers (arrays of length n) in ceil(regions / 48) launches: the whole network behind one optimiser step. */
int da_conv3d_k3_prepack_many(int n, const float* const* w_tio, const int* C1, const int* C2, const int* Cout, const int* dgrad,
                              const int* N, const int* D, const int* H, const int* W,
                              void* const* b0, const size_t* n0, void* const* b1, const size_t* n1, int* used, void* stream);

/* The same for the kernel families OUTSIDE the split matrix kernels -- the folded up-sampling convolution (da_upconv3d_k3_fwd / _dgrad; voxel_morph.py:72-80 + the conv
 * behind it), the native stride-2 convolution (modules.py:48, stride 2), the flow convolution (voxel_morph.py:57) and the first layers' thin kernels -- whose per-call
 * packs were 22 launches of 5 - 20 us in the dependent chain of a registration step.
 *   da_conv3d_k3_prepack_any        the family that da_conv3d_k3_fwd / _dgrad (up2 = 0; stride 1 | 2) or da_upconv3d_k3_fwd / _dgrad (up2 = 1; D, H, W = the coarse
 *                                   extents) runs for this shape packs into buf (cap bytes) and launches nothing else.  *need: bytes of such a pack (0: the shape keeps
 *                                   nothing here -- split matrix kernels: da_conv3d_k3_prepack; direct kernels: no pack); buf NULL: size query.  *tag: (family,
 *                                   direction), to be passed back; *filled: buf was written.  ws: the call's usual workspace.
 *   da_conv3d_k3_use_prepacked_any  the NEXT of those calls on `w_tio` whose family / direction carries `tag` reads its packed operand from buf (one call only). */
int da_conv3d_k3_prepack_any(const float* w_tio, int C1, int C2, int Cout, int dgrad, int stride, int up2, int N, int D, int H, int W,
                             void* buf, size_t cap, size_t* need, int* tag, int* filled, void* ws, size_t ws_bytes, void* stream);
void da_conv3d_k3_use_prepacked_any(const float* w_tio, const void* buf, size_t cap, int tag);

/* ---- NCC loss (row a12; lib/loss.py:493-501) -------------------------------------------------- */
size_t da_ncc_ws_bytes(int N, long long V);
int da_ncc_fwd(const float* x, const float* y, int N, long long V, float* loss, double* stats /*[N][8]*/,
               void* ws, size_t ws_bytes, void* stream);
int da_ncc_bwd(const float* x, const float* y, const double* stats, const float* dloss,
               float* dx, float* dy, int N, long long V, void* stream);

/* ---- bending-energy loss (row a13; lib/loss.py:687-730) --------------------------------------- */
size_t da_bending_ws_bytes(int N, int D, int H, int W);
/* norm: 2 = 'L2' (the weighted squared differences, :721-727); 1 = any other value of the reference's `norm`: the block at :721-727
 * is skipped and the loss is the plain mean of the |differences| (:729). */
int da_bending_fwd(const float* disp, int N, int D, int H, int W, const float* spacing3, int normalize, int norm,
                   float* loss, void* ws, size_t ws_bytes, void* stream);
int da_bending_bwd(const float* disp, const float* dloss, float* d_disp, int N, int D, int H, int W,
                   const float* spacing3, int normalize, int norm, void* stream);

/* ---- cross-entropy family of the loss registry (lib/loss.py:739-761) ------------------------------------------------------
 * logits [M][C] (channels-last voxels), C <= 64.  mode 0: 'cross_entropy' = torch.nn.CrossEntropyLoss(ignore_index, reduction);
 * mode 1: 'focal' = FocalLoss.forward (lib/loss.py:181-213): -alpha[t] (1 + P[t])^gamma log_softmax(x)[t] with P = softmax(x) when
 * `softmax` else x -- the reference's `1 - F.nll_loss(P, t)` is 1 PLUS p_t, kept; mode 2: 'soft_cross_entropy' = SoftCrossEntropy.forward
 * (:115-154) with a class-probability target [M][C]: mean over voxels of sum_c -t_c log_softmax(x)_c (softmax) or -t_c log(max(x_c, 1e-8)).
 * reduction: 0 mean (cross_entropy: over the non-ignored voxels), 1 sum.  denom (1 float, device) is written by fwd and read by bwd. */
size_t da_xent_ws_bytes(void);
int da_xent_fwd(const float* logits, const void* labels, int label_bytes, const float* soft_target, const float* alpha,
                long long M, int C, int mode, int softmax, float gamma, long long ignore_index, int reduction,
                float* loss, float* denom, void* ws, size_t ws_bytes, void* stream);
int da_xent_bwd(const float* logits, const void* labels, int label_bytes, const float* soft_target, const float* alpha,
                const float* dloss, const float* denom, float* dlogits, long long M, int C, int mode, int softmax, float gamma,
                long long ignore_index, void* stream);

/* ---- UNet_generator options (unets.py:230-237) ---------------------------------------------------------------
 * maxpool=False: nn.Conv3d(k2, s2, p0) down-sampler = the adjoint of the k2/s2 transposed conv (same pointwise MFMA kernels).
 * D, H, W are the COARSE (output) dims, the input is 2D x 2H x 2W; w_tio [8][Cin][Cout]; dw_toi [8][Cout][Cin]. */
int da_conv_k2s2_fwd(const float* x, const float* w_tio, const float* bias, float* y,
                     int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int da_conv_k2s2_dgrad(const float* dy, const float* w_tio, float* dx,
                       int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
size_t da_conv_k2s2_wgrad_ws_bytes(int N, int D, int H, int W, int Cin, int Cout);
int da_conv_k2s2_wgrad(const float* x, const float* dy, float* dw_toi, float* dbias,
                       int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* upsample=True: nn.Upsample(scale_factor=2, mode='trilinear') (align_corners=False); x [N][D][H][W][C] -> y [N][2D][2H][2W][C] */
int da_upsample_trilinear2_fwd(const float* x, float* y, int N, int D, int H, int W, int C, void* stream);
int da_upsample_trilinear2_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C, void* stream);

/* ---- device data path (lib/transforms.py:71-92 SitkToTensor, :124-158 CropTensor, :508-649 Partition) ------
 * src_dtype: 0 f32, 1 f64, 2 i16, 3 u8, 4 i32.  Volumes are [D][H][W] (numpy order z, y, x), tile3 / overlap3 likewise. */
int da_clamp01_to_f32(const void* src, int src_dtype, float* dst, long long n, void* stream);
int da_crop3d(const void* src, void* dst, int elem_bytes, long long C, int D, int H, int W,
              int d0, int h0, int w0, int Do, int Ho, int Wo, void* stream);
/* overlap tiling with numpy 'reflect' padding; tiles [ceil(D/ez)*ceil(H/ey)*ceil(W/ex)][tz][ty][tx], e = tile - 2*overlap */
int da_partition_tiles(const void* vol, void* tiles, int elem_bytes, int D, int H, int W, const int* tile3, const int* overlap3, void* stream);
/* vote == 0: copy the effective core of each tile; vote != 0 (uint8 labels): per-voxel majority over the covering tiles */
int da_assemble_tiles(const void* tiles, void* vol, int elem_bytes, int D, int H, int W, const int* tile3, const int* overlap3, int vote, void* stream);

/* synthetic volumes generated on the device (row f4; the structure of the BASELINE configs' synthetic data, the reference's sample
 * convention lib/datasets.py:150-166: image [N][D][H][W] fp32 in [0,1], segmentation uint8).  Counter-based hash: element = f(seed,
 * sample0 + n, voxel), restated bit-exactly in oracle/datapath.py.  mode 0: iid image ~ U[0,1), labels ~ U{0..C-1};
 * mode 1: blocky coordinate-function labels, image = clamp(label / (C-1) + noise U, 0, 1).  img or labels may be NULL. */
int da_synth_volume(float* img, unsigned char* labels, int N, int D, int H, int W, int n_classes, int mode, float noise,
                    unsigned int seed, int sample0, void* stream);

/* ---- LNCC similarity (lib/loss.py:589-617 VoxelMorphLNCC = registry 'lncc', and :512-586 LNCCLoss) -----
 * I, J: [N][D][H][W] fp32 (single channel); all-ones F^3 window with dilation `dil` and stride `stride` (1, 1 for VoxelMorphLNCC),
 * valid padding; loss = 1 - mean(cross^2 / (Ivar Jvar + eps)).  Output extent per axis: (L - dil (F-1) - 1) / stride + 1.
 * sums: [5][N][Do][Ho][Wo] floats written by fwd and consumed by bwd of the SAME geometry, opaque to the caller: the five window
 * sums (I, J, I^2, J^2, IJ) in the separable form; in the z-marching form (dil = stride = 1, F = 5 or 9: one fused kernel per
 * direction, reglosses.hip) the five per-window backward terms A', B', C', E1', E2'.  DA_LNCC_MARCH=0 keeps the separable form. */
size_t da_lncc_ws_bytes(int N, int D, int H, int W, int F, int dil, int stride);
int da_lncc_fwd(const float* I, const float* J, int N, int D, int H, int W, int F, int dil, int stride, float eps,
                float* loss, float* sums, void* ws, size_t ws_bytes, void* stream);
int da_lncc_bwd(const float* I, const float* J, const float* sums, const float* dloss, float* dI, float* dJ,
                int N, int D, int H, int W, int F, int dil, int stride, float eps, void* ws, size_t ws_bytes, void* stream);

/* ---- displacement gradient regulariser (row f2; lib/loss.py:625-671 gradientLoss, registry 'gradient') ------------------
 * disp [N][D][H][W][3]; norm = 2 ('L2') or 1; keeps the reference's +/- quirk along H and W (loss.py:661,663). */
size_t da_gradloss_ws_bytes(int N, int D, int H, int W);
int da_gradloss_fwd(const float* disp, int N, int D, int H, int W, const float* spacing3, int normalize, int norm,
                    float* loss, void* ws, size_t ws_bytes, void* stream);
int da_gradloss_bwd(const float* disp, const float* dloss, float* d_disp, int N, int D, int H, int W,
                    const float* spacing3, int normalize, int norm, void* stream);

/* ---- eval: argmax + per-class overlap counts (row a15; models/segmentation.py:188-194) -------- */
/* counts[N][C][3] uint64 = (|pred==c|, |truth==c|, |pred==c & truth==c|), must be zero-filled; pred (may be NULL) uint8 [N][V]. */
int da_argmax_dice_counts(const float* logits, const void* truth, int label_bytes, int N, long long V, int C,
                          unsigned long long* counts, unsigned char* pred, void* stream);

/* ---- eval on label maps (lib/evalMetrics.py:103-217 get_multi_metric / cal_metric / get_multiclass_dice,
 *      lib/loss.py:348-391 DiceLossOnLabel): every one of those metrics is a function of these integer counts ---------- */
/* pred / truth: uint8 (1) or int64 (8) label maps [N][V]; labels outside [0, C) are ignored; counts as above, zero-filled. */
int da_label_overlap_counts(const void* pred, int pred_bytes, const void* truth, int truth_bytes, int N, long long V, int C,
                            unsigned long long* counts, void* stream);

/* ---- optimiser (models/segmentation.py:91 torch.optim.Adam defaults) -------------------------- */
int da_adam_step(float* p, const float* g, float* m, float* v, long long n,
                 float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream);

/* the same update with its per-step scalars in DEVICE memory -- state6 = {lr / bc1, beta1, beta2, eps, 1 / sqrt(bc2), grad_scale} -- so
 * that the launch can be captured in a HIP graph and replayed (deepatlas_amd/graphs.py); da_adam_host_state fills a HOST array with
 * those six values for step `step`, using da_adam_step's own expressions (the two paths update bit-identically). */
int da_adam_host_state(float lr, float beta1, float beta2, float eps, int step, float grad_scale, float* state6_host);
int da_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float* state6, void* stream);

/* =====================================================================================================================
 * bf16 ACTIVATION STORAGE (BASELINE.json configs[4]: "bf16 mixed precision"; SURVEY.md 8(d) config 5: "bf16 activations / MFMA, fp32
 * master weights, fp32 reductions").  The reference has no such mode (train_seg.py is fp32-only); these are the `_bf16` twins of the
 * entries above for a network whose INTERNAL activation and gradient tensors are stored as bf16 (same NDHWC layout, 2 bytes per element).
 * `void*` arguments are those bf16 tensors; every `float*` keeps its meaning (weights, biases, statistics, scale / shift, weight and bias
 * gradients, logits and their gradient, losses).  Arithmetic is fp32 (double in the reductions); a value is rounded to nearest-even once,
 * when it is stored.  BatchNorm statistics fused into a producer's epilogue are those of the fp32 values before that rounding.
 * The convolution twins need the bf16 matrix mode (da_set_matrix_mode(1)) and the matrix-core shapes; DA_ERR_UNSUPPORTED otherwise -- the
 * caller then converts with da_cast_* around the fp32 entry (deepatlas_amd/ops.py: call_act).
 * ===================================================================================================================== */
int da_cast_f32_to_bf16(const float* x, void* y, long long numel, void* stream);
int da_cast_bf16_to_f32(const void* x, float* y, long long numel, void* stream);
/* norm_act.hip */
int da_bn_train_stats_bf16(const void* x, long long M, int C, const float* gamma, const float* beta,
                           float eps, float momentum, float* running_mean, float* running_var,
                           float* mean, float* rstd, float* scale, float* shift, void* ws, size_t ws_bytes, void* stream);
int da_bn_act_fwd_bf16(const void* x, const float* scale, const float* shift, float act_slope, void* y, long long M, int C, void* stream);
int da_bn_act_bwd_dbias_bf16(const void* dy, const void* x, const float* mean, const float* rstd,
                             const float* scale, const float* shift, float act_slope, int train,
                             void* dx, float* dgamma, float* dbeta, float* dxsum, long long M, int C,
                             void* ws, size_t ws_bytes, void* stream);
int da_act_bwd_bf16(const void* dy, const void* y, float act_slope, void* dx, long long numel, void* stream);
int da_act_bwd_add_dbias_bf16(const void* g1, const void* g2, const void* y, float act_slope, void* dx, float* dbias,
                              long long M, int C, void* ws, size_t ws_bytes, void* stream);
int da_act_bwd_add_partial_bf16(const void* g1, const void* g2, const void* y, float act_slope, void* dx,
                                long long M, int C, void* partial, size_t partial_bytes, int* nparts, void* stream);
int da_colsum_bf16(const void* x, long long M, int C, float* out, void* ws, size_t ws_bytes, void* stream);
/* pool.hip */
int da_maxpool2_fwd_bf16(const void* x, void* y, int N, int D, int H, int W, int C, void* stream);
int da_maxpool2_fwd_pro_bf16(const void* x, const float* pro_scale, const float* pro_shift, float pro_slope, void* act, void* y,
                             int N, int D, int H, int W, int C, void* stream);      /* the maximum is taken over the ROUNDED activations */
int da_maxpool2_bwd_bf16(const void* dy, const void* x, void* dx, int N, int D, int H, int W, int C, void* stream);
int da_maxpool2_bwd_add_bf16(const void* dy, const void* x, const void* gskip, void* dx, int N, int D, int H, int W, int C, void* stream);
int da_upsample_nearest_fwd_bf16(const void* x, void* y, int N, int D, int H, int W, int C, int Do, int Ho, int Wo, void* stream);
int da_upsample_nearest_bwd_bf16(const void* dy, void* dx, int N, int D, int H, int W, int C, int Do, int Ho, int Wo, void* stream);
/* conv3d.hip / conv3d_mfma.hip: `bf16_mask` = one bit per activation argument in signature order (set = bf16).  Native: all set. */
int da_conv3d_k3_fwd_bf16(const void* in1, int C1, const void* in2, int C2, const float* w_tio, const float* bias, void* out,
                          int N, int D, int H, int W, int Cout, int stride, float act_slope,
                          void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask);
int da_conv3d_k3_fwd_bnstats_bf16(const void* in1, int C1, const void* in2, int C2, const float* w_tio, const float* bias, void* out,
                                  int N, int D, int H, int W, int Cout, int stride,
                                  double* stats_partial, int stats_capacity, int* stats_nparts,
                                  void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask);
int da_conv3d_k3_fwd_pro_bf16(const void* in1, int C1, const float* pro1_scale, const float* pro1_shift, float pro1_slope,
                              const void* in2, int C2, const float* pro2_scale, const float* pro2_shift, float pro2_slope,
                              const float* w_tio, const float* bias, void* out,
                              int N, int D, int H, int W, int Cout, float act_slope,
                              double* stats_partial, int stats_capacity, int* stats_nparts,
                              void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask);
int da_conv3d_k3_wgrad_pro_bf16(const void* in1, int C1, const float* pro1_scale, const float* pro1_shift, float pro1_slope,
                                const void* in2, int C2, const float* pro2_scale, const float* pro2_shift, float pro2_slope,
                                const void* dy, float* dw_tio, int N, int D, int H, int W, int Cout,
                                void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask);
int da_conv3d_k3_dgrad_bf16(const void* dy, const float* w_tio, void* dx1, int C1, void* dx2, int C2,
                            int N, int D, int H, int W, int Cout, int stride, void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask);
int da_conv3d_k3_wgrad_bf16(const void* in1, int C1, const void* in2, int C2, const void* dy, float* dw_tio, float* dbias,
                            int N, int D, int H, int W, int Cout, int stride, void* ws, size_t ws_bytes, void* stream, unsigned bf16_mask);
/* deconv.hip / pointwise_mfma.hip: transposed conv k2 s2 (input, output, their gradients bf16); 1x1x1 head (input and its gradient
 * bf16; logits and their gradient fp32) */
int da_deconv_k2s2_fwd_bf16(const void* in, const float* w_tio, const float* bias, void* out,
                            int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int da_deconv_k2s2_fwd_bnstats_bf16(const void* in, const float* w_tio, const float* bias, void* out,
                                    int N, int D, int H, int W, int Cin, int Cout,
                                    double* stats_partial, int stats_capacity, int* stats_nparts, void* ws, size_t ws_bytes, void* stream);
int da_deconv_k2s2_dgrad_bf16(const void* dy, const float* w_tio, void* dx,
                              int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int da_deconv_k2s2_wgrad_bf16(const void* in, const void* dy, float* dw_tio, float* dbias,
                              int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int da_conv1x1_fwd_bf16(const void* in, const float* w_io, const float* bias, float* out,
                        long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int da_conv1x1_fwd_pro_bf16(const void* in, const float* pro_scale, const float* pro_shift, float pro_slope,
                            const float* w_io, const float* bias, float* out, long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int da_conv1x1_dgrad_bf16(const float* dy, const float* w_io, void* dx, long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int da_conv1x1_wgrad_bf16(const void* in, const float* dy, float* dw_io, float* dbias,
                          long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int da_conv1x1_wgrad_pro_bf16(const void* in, const float* pro_scale, const float* pro_shift, float pro_slope,
                              const float* dy, float* dw_io, float* dbias, long long M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* headdice.hip: fused head + softmax + Dice with a bf16 input x and a bf16 gradient dx */
int da_head_dice_fwd_bf16(const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                          const float* w_io, const float* bias, const void* labels, int label_bytes,
                          int N, long long V, int Cin, int C, int weight_type, int no_bg, float eps,
                          float* loss, float* coef, void* ws, size_t ws_bytes, void* stream);
int da_head_dice_bwd_bf16(const void* x, const float* pro_scale, const float* pro_shift, float pro_slope,
                          const float* w_io, const float* bias, const void* labels, int label_bytes,
                          const float* coef, const float* dloss, void* dx, float* dw_io, float* dbias,
                          int N, long long V, int Cin, int C, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPATLAS_HIP_H */
