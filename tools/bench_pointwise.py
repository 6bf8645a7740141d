#!/usr/bin/env python
"""Micro-benchmark the pointwise MFMA GEMMs through the C ABI: transposed conv k2 s2 (fwd / fwd + BN statistics / dgrad / wgrad) and the
1x1x1 head (fwd / dgrad / wgrad), with their algorithmic HBM bytes.  Usage: python tools/bench_pointwise.py [--iters 10]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepatlas_amd import _native as nat
from deepatlas_amd._native import call, ptr, stream, workspace


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--only', default='', help='run only the cases whose name contains this')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    st = stream()
    # up-sampler 32 -> 32, coarse 2 x 80 x 96 x 80 (fine = the bench's full resolution)
    N, D, H, W, Ci, Co = 2, 80, 96, 80, 32, 32
    x = torch.rand((N, D, H, W, Ci), device=dev) - 0.5
    w = torch.rand((8, Ci, Co), device=dev) - 0.5
    y = torch.empty((N, 2 * D, 2 * H, 2 * W, Co), device=dev)
    dy = torch.rand_like(y) - 0.5
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    nblk = (N * D * H * W + 255) // 256
    pb = torch.empty((nblk, 2, Co), dtype=torch.float64, device=dev)
    npar = ctypes.c_int(0)
    wp, wn = workspace.get(max(nat.lib().da_pointwise_ws_bytes(8, Ci, Co), nat.lib().da_deconv_k2s2_wgrad_ws_bytes(N, D, H, W, Ci, Co)), dev)
    fine, coarse = y.numel() * 4, x.numel() * 4
    cases = [
        ('deconv fwd', lambda: call('da_deconv_k2s2_fwd', ptr(x), ptr(w), None, ptr(y), N, D, H, W, Ci, Co, wp, wn, st), fine + coarse),
        ('deconv fwd+stats', lambda: call('da_deconv_k2s2_fwd_bnstats', ptr(x), ptr(w), None, ptr(y), N, D, H, W, Ci, Co, ptr(pb), nblk, ctypes.byref(npar), wp, wn, st), fine + coarse),
        ('deconv dgrad', lambda: call('da_deconv_k2s2_dgrad', ptr(dy), ptr(w), ptr(dx), N, D, H, W, Ci, Co, wp, wn, st), fine + coarse),
        ('deconv wgrad', lambda: call('da_deconv_k2s2_wgrad', ptr(x), ptr(dy), ptr(dw), None, N, D, H, W, Ci, Co, wp, wn, st), fine + coarse),
    ]
    # the up-sampler block's fused backward (BatchNorm-backward sums pass + apply-on-the-fly + data / weight / bias gradients)
    mean, rstd = torch.rand(Co, device=dev) - 0.5, torch.rand(Co, device=dev) + 0.5
    gam, bet = torch.rand(Co, device=dev) + 0.5, torch.rand(Co, device=dev) - 0.5
    dbias, dgam, dbet = torch.empty(Co, device=dev), torch.empty(Co, device=dev), torch.empty(Co, device=dev)
    fb = nat.lib().da_deconv_k2s2_bn_bwd_ws_bytes(N, D, H, W, Ci, Co)
    wpf, wnf = workspace.get(fb, dev)
    yraw = torch.rand_like(y) - 0.5
    cases.append(('deconv bn bwd (fused)', lambda: call('da_deconv_k2s2_bn_bwd', ptr(dy), ptr(yraw), ptr(mean), ptr(rstd), ptr(gam), ptr(bet), 0.01, ptr(x), ptr(w), ptr(dx), ptr(dw),
                                                       ptr(dbias), ptr(dgam), ptr(dbet), N, D, H, W, Ci, Co, None, 0, wpf, wnf, st), 2 * fine + 2 * coarse))
    # head 16 -> 32 at full resolution
    M, Hi, Ho = 2 * 160 * 192 * 160, 16, 32
    hx = torch.rand((M, Hi), device=dev) - 0.5
    hw = torch.rand((Hi, Ho), device=dev) - 0.5
    hy = torch.empty((M, Ho), device=dev)
    hdy = torch.rand_like(hy) - 0.5
    hdx = torch.empty_like(hx)
    hdw = torch.empty_like(hw)
    sc, sh = torch.rand(Hi, device=dev) + 0.5, torch.rand(Hi, device=dev) - 0.5
    wp2, wn2 = workspace.get(max(nat.lib().da_pointwise_ws_bytes(1, Hi, Ho), nat.lib().da_conv1x1_wgrad_ws_bytes(M, Hi, Ho)), dev)
    bi, bo = hx.numel() * 4, hy.numel() * 4
    cases += [
        ('head fwd', lambda: call('da_conv1x1_fwd', ptr(hx), ptr(hw), None, ptr(hy), M, Hi, Ho, wp2, wn2, st), bi + bo),
        ('head fwd (prologue)', lambda: call('da_conv1x1_fwd_pro', ptr(hx), ptr(sc), ptr(sh), 0.01, ptr(hw), None, ptr(hy), M, Hi, Ho, wp2, wn2, st), bi + bo),
        ('head dgrad', lambda: call('da_conv1x1_dgrad', ptr(hdy), ptr(hw), ptr(hdx), M, Hi, Ho, wp2, wn2, st), bi + bo),
        ('head wgrad', lambda: call('da_conv1x1_wgrad', ptr(hx), ptr(hdy), ptr(hdw), None, M, Hi, Ho, wp2, wn2, st), bi + bo),
        ('head wgrad (prologue)', lambda: call('da_conv1x1_wgrad_pro', ptr(hx), ptr(sc), ptr(sh), 0.01, ptr(hdy), ptr(hdw), None, M, Hi, Ho, wp2, wn2, st), bi + bo),
    ]
    for name, fn, nbytes in cases:
        if a.only and a.only not in name:
            continue
        ms = timeit(fn, a.iters)
        print('%-24s %7.3f ms  %7.1f MB  %7.1f GB/s  %.3f of HBM peak' % (name, ms, nbytes / 1e6, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000.0))


if __name__ == '__main__':
    main()
