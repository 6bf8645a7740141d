/* oracle/prim.c -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product path).
 *
 * Plain-C restatement, from their published definitions, of the primitive ops the DeepAtlas hot path is made of, with
 * double accumulation, as an ATen-independent cross-check of the torch-CPU oracle (oracle/nets.py, oracle/losses.py).
 * The reference calls these through PyTorch (third party, unpinned: requirements.txt:1); call sites:
 *   conv 3x3x3 pad 1 stride s      lib/network_factory/unets.py:30,36 ; modules.py:48 ; voxel_morph.py:57
 *   ConvTranspose3d k2 s2          lib/network_factory/unets.py:49,55
 *   BatchNorm3d (training) + LeakyReLU   unets.py:31-32,51-52
 *   MaxPool3d(2)                   unets.py:230
 *   F.interpolate(nearest)         voxel_morph.py:72-80
 *   F.grid_sample (trilinear, zeros, align_corners=True)   voxel_morph.py:91
 *   F.softmax(dim=1)               lib/loss.py:431
 * Layout: the reference's own NCDHW, dense fp32.  tests/test_oracle_prim.py pins this file against torch-CPU and against the
 * golden vectors generated from the reference (tests/golden/ops.npz).
 */
#include <math.h>
#include <stddef.h>

#define IDX5(n, c, d, h, w, C, D, H, W) (((((size_t)(n) * (C) + (c)) * (D) + (d)) * (H) + (h)) * (W) + (w))

/* y[n][co][od][oh][ow] = b[co] + sum_{ci,kd,kh,kw} x[n][ci][od*s-1+kd][oh*s-1+kh][ow*s-1+kw] * w[co][ci][kd][kh][kw] */
void prim_conv3d_k3(const float* x, const float* w, const float* b, float* y,
                    int N, int Cin, int D, int H, int W, int Cout, int stride) {
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int od = 0; od < Do; ++od)
                for (int oh = 0; oh < Ho; ++oh)
                    for (int ow = 0; ow < Wo; ++ow) {
                        double acc = b ? (double)b[co] : 0.0;
                        for (int ci = 0; ci < Cin; ++ci)
                            for (int kd = 0; kd < 3; ++kd) {
                                const int d = od * stride - 1 + kd;
                                if (d < 0 || d >= D) continue;
                                for (int kh = 0; kh < 3; ++kh) {
                                    const int h = oh * stride - 1 + kh;
                                    if (h < 0 || h >= H) continue;
                                    for (int kw = 0; kw < 3; ++kw) {
                                        const int ww = ow * stride - 1 + kw;
                                        if (ww < 0 || ww >= W) continue;
                                        acc += (double)x[IDX5(n, ci, d, h, ww, Cin, D, H, W)] *
                                               (double)w[((((size_t)co * Cin + ci) * 3 + kd) * 3 + kh) * 3 + kw];
                                    }
                                }
                            }
                        y[IDX5(n, co, od, oh, ow, Cout, Do, Ho, Wo)] = (float)acc;
                    }
}

/* ConvTranspose3d kernel 2 stride 2: y[n][co][2d+kd][2h+kh][2w+kw] = b[co] + sum_ci x[n][ci][d][h][w] * w[ci][co][kd][kh][kw] */
void prim_deconv_k2s2(const float* x, const float* w, const float* b, float* y, int N, int Cin, int D, int H, int W, int Cout) {
    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int od = 0; od < Do; ++od)
                for (int oh = 0; oh < Ho; ++oh)
                    for (int ow = 0; ow < Wo; ++ow) {
                        const int d = od / 2, kd = od % 2, h = oh / 2, kh = oh % 2, ww = ow / 2, kw = ow % 2;
                        double acc = b ? (double)b[co] : 0.0;
                        for (int ci = 0; ci < Cin; ++ci)
                            acc += (double)x[IDX5(n, ci, d, h, ww, Cin, D, H, W)] *
                                   (double)w[((((size_t)ci * Cout + co) * 2 + kd) * 2 + kh) * 2 + kw];
                        y[IDX5(n, co, od, oh, ow, Cout, Do, Ho, Wo)] = (float)acc;
                    }
}

/* BatchNorm (training): per-channel mean / biased variance over (N, D, H, W); y = (x - mean) / sqrt(var + eps) * g + b, then
 * LeakyReLU(slope) when slope >= 0.  mean_out / var_out (biased) are optional. */
void prim_bn_train_act(const float* x, const float* g, const float* b, float* y, double* mean_out, double* var_out,
                       int N, int C, long long S, double eps, double slope) {
    for (int c = 0; c < C; ++c) {
        double s1 = 0.0;
        for (int n = 0; n < N; ++n) {
            const float* p = x + ((size_t)n * C + c) * S;
            for (long long i = 0; i < S; ++i) s1 += p[i];
        }
        const double mean = s1 / ((double)N * S);
        double s2 = 0.0;
        for (int n = 0; n < N; ++n) {
            const float* p = x + ((size_t)n * C + c) * S;
            for (long long i = 0; i < S; ++i) { const double d = p[i] - mean; s2 += d * d; }
        }
        const double var = s2 / ((double)N * S);
        if (mean_out) mean_out[c] = mean;
        if (var_out) var_out[c] = var;
        const double inv = 1.0 / sqrt(var + eps);
        for (int n = 0; n < N; ++n) {
            const float* p = x + ((size_t)n * C + c) * S;
            float* q = y + ((size_t)n * C + c) * S;
            for (long long i = 0; i < S; ++i) {
                double v = (p[i] - mean) * inv * (g ? g[c] : 1.0) + (b ? b[c] : 0.0);
                if (slope >= 0.0 && v < 0.0) v *= slope;
                q[i] = (float)v;
            }
        }
    }
}

/* MaxPool3d(2): floor output size, first maximum wins (index order d, h, w); idx (optional) = flat input offset in the plane */
void prim_maxpool2(const float* x, float* y, long long* idx, int NC, int D, int H, int W) {
    const int Do = D / 2, Ho = H / 2, Wo = W / 2;
    for (int nc = 0; nc < NC; ++nc)
        for (int od = 0; od < Do; ++od)
            for (int oh = 0; oh < Ho; ++oh)
                for (int ow = 0; ow < Wo; ++ow) {
                    float best = -INFINITY; long long bi = -1;
                    for (int kd = 0; kd < 2; ++kd)
                        for (int kh = 0; kh < 2; ++kh)
                            for (int kw = 0; kw < 2; ++kw) {
                                const long long off = ((long long)(2 * od + kd) * H + (2 * oh + kh)) * W + (2 * ow + kw);
                                const float v = x[(size_t)nc * D * H * W + off];
                                if (bi < 0 || v > best) { best = v; bi = off; }
                            }
                    const size_t o = (((size_t)nc * Do + od) * Ho + oh) * Wo + ow;
                    y[o] = best;
                    if (idx) idx[o] = bi;
                }
}

/* F.interpolate(mode='nearest', size=(Do,Ho,Wo)): src = min(floor(dst * (in / out)), in - 1), the scale in float */
void prim_upsample_nearest(const float* x, float* y, int NC, int D, int H, int W, int Do, int Ho, int Wo) {
    const float sd = (float)D / (float)Do, sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
    for (int nc = 0; nc < NC; ++nc)
        for (int od = 0; od < Do; ++od) {
            int d = (int)floorf((float)od * sd); if (d > D - 1) d = D - 1;
            for (int oh = 0; oh < Ho; ++oh) {
                int h = (int)floorf((float)oh * sh); if (h > H - 1) h = H - 1;
                for (int ow = 0; ow < Wo; ++ow) {
                    int w = (int)floorf((float)ow * sw); if (w > W - 1) w = W - 1;
                    y[(((size_t)nc * Do + od) * Ho + oh) * Wo + ow] = x[(((size_t)nc * D + d) * H + h) * W + w];
                }
            }
        }
}

/* F.grid_sample(src, grid, mode='bilinear', padding_mode='zeros', align_corners=True) in 3-D.
 * grid[n][d][h][w][0..2] = (x, y, z) in [-1, 1]; pixel = (g + 1) / 2 * (size - 1); out-of-volume corners contribute zero. */
void prim_grid_sample3d(const float* src, const float* grid, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo) {
    for (int n = 0; n < N; ++n)
        for (int od = 0; od < Do; ++od)
            for (int oh = 0; oh < Ho; ++oh)
                for (int ow = 0; ow < Wo; ++ow) {
                    const float* g = grid + ((((size_t)n * Do + od) * Ho + oh) * Wo + ow) * 3;
                    const double px = ((double)g[0] + 1.0) * 0.5 * (W - 1), py = ((double)g[1] + 1.0) * 0.5 * (H - 1),
                                 pz = ((double)g[2] + 1.0) * 0.5 * (D - 1);
                    const double fx = floor(px), fy = floor(py), fz = floor(pz);
                    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
                    const double tx = px - fx, ty = py - fy, tz = pz - fz;
                    for (int c = 0; c < C; ++c) {
                        double acc = 0.0;
                        for (int k = 0; k < 8; ++k) {
                            const int cx = k & 1, cy = (k >> 1) & 1, cz = k >> 2;
                            const int x = x0 + cx, y = y0 + cy, z = z0 + cz;
                            if (x < 0 || x >= W || y < 0 || y >= H || z < 0 || z >= D) continue;
                            const double wgt = (cx ? tx : 1.0 - tx) * (cy ? ty : 1.0 - ty) * (cz ? tz : 1.0 - tz);
                            acc += wgt * (double)src[IDX5(n, c, z, y, x, C, D, H, W)];
                        }
                        out[IDX5(n, c, od, oh, ow, C, Do, Ho, Wo)] = (float)acc;
                    }
                }
}

/* softmax over the channel axis of an N x C x S tensor */
void prim_softmax_c(const float* x, float* y, int N, int C, long long S) {
    for (int n = 0; n < N; ++n)
        for (long long i = 0; i < S; ++i) {
            double m = -INFINITY;
            for (int c = 0; c < C; ++c) { const double v = x[((size_t)n * C + c) * S + i]; if (v > m) m = v; }
            double sum = 0.0;
            for (int c = 0; c < C; ++c) sum += exp((double)x[((size_t)n * C + c) * S + i] - m);
            for (int c = 0; c < C; ++c) y[((size_t)n * C + c) * S + i] = (float)(exp((double)x[((size_t)n * C + c) * S + i] - m) / sum);
        }
}
