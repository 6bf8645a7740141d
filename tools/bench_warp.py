#!/usr/bin/env python
"""Micro-benchmark of the gather kernels of the joint step through the C ABI at one 160x192x160 sample: image / 32-channel warp, the fused
label-warp Dice, the label adjoint scatter and the logit-gradient pass, on a smooth (registration-like) and on a per-voxel random displacement.
python tools/bench_warp.py [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepatlas_amd import _native as nat
from deepatlas_amd._native import call, ptr, stream, workspace
from deepatlas_amd.lib.datasets import structured_labels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--shape', type=int, nargs=3, default=[160, 192, 160])
    a = ap.parse_args()
    D, H, W = a.shape
    V, C = D * H * W, 32
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    zz, yy, xx = torch.meshgrid(torch.linspace(0, 6.28, D), torch.linspace(0, 6.28, H), torch.linspace(0, 6.28, W), indexing='ij')
    smooth = (torch.stack([torch.sin(zz + yy), torch.cos(xx - zz), torch.sin(xx + 2 * yy)], -1) * 0.03)[None].contiguous().to(dev)
    rnd = ((torch.rand((1, D, H, W, 3), generator=g) - 0.5) * 0.05).to(dev)
    img = torch.rand((1, D, H, W, 1), generator=g).to(dev)
    src32 = torch.rand((1, D, H, W, C), generator=g).to(dev)
    out1, out32 = torch.empty_like(img), torch.empty_like(src32)
    deform = torch.empty_like(rnd)
    lab_m = structured_labels((D, H, W), C).to(torch.uint8).to(dev).reshape(1, -1).contiguous()
    lab_t = structured_labels((D, H, W), C, seed=3).to(torch.uint8).to(dev).reshape(1, -1).contiguous()
    loss = torch.empty(1, device=dev); coef = torch.empty((2, 1, C), device=dev)
    wsb = nat.lib().da_label_warp_dice_ws_bytes(1, C)
    wp, wn = workspace.get(wsb, dev)
    B = torch.empty((1, V, C), device=dev)
    st = stream()
    gl = torch.ones(1, device=dev)
    d_disp = torch.empty_like(rnd)
    prob = torch.softmax(torch.randn((1, V, C), generator=g), -1).to(dev)
    dlog = torch.empty_like(prob)
    coef_a = torch.rand((2, 1, C), generator=g).to(dev)
    flush = torch.empty(1 << 28, device=dev)

    def timed(name, fn, nbytes):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        # cold: a 1 GiB fill between calls evicts L2 and the 256 MB Infinity Cache, the state the call sees inside a training step
        cold = []
        for _ in range(min(a.iters, 8)):
            flush.fill_(1.0)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record(); fn(); c1.record()
            torch.cuda.synchronize()
            cold.append(c0.elapsed_time(c1))
        cms = sorted(cold)[len(cold) // 2]
        print('%-40s hot %7.3f ms %7.1f GB/s %.3f of peak | cold %7.3f ms %7.1f GB/s %.3f of peak' % (
            name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000.0, cms, nbytes / cms / 1e6, nbytes / cms / 1e6 / 8000.0))

    tiny = (torch.randn((1, D, H, W, 3), generator=g) * 1e-6).to(dev)       # an untrained flow head: sample points within 1e-4 voxel of the grid
    for dn, u in (('smooth', smooth), ('random', rnd), ('tiny', tiny)):
        timed('da_warp_fwd[1] %s' % dn, lambda: call('da_warp_fwd', ptr(img), ptr(u), ptr(deform), ptr(out1), 1, D, H, W, 1, st), 5 * 4 * V)
        timed('da_warp_bwd[1] d_disp only %s' % dn, lambda: call('da_warp_bwd', ptr(out1), ptr(img), ptr(u), ptr(d_disp), None, 1, D, H, W, 1, st), 8 * 4 * V)
        timed('da_warp_fwd[32] %s' % dn, lambda: call('da_warp_fwd', ptr(src32), ptr(u), ptr(deform), ptr(out32), 1, D, H, W, C, st), 67 * 4 * V)
        timed('da_label_warp_dice_fwd %s' % dn, lambda: call('da_label_warp_dice_fwd', ptr(lab_m), 1, ptr(lab_t), 1, ptr(u), 1, D, H, W, C, 0, 0, 1e-6,
                                                            ptr(loss), ptr(coef), wp, wn, st), 14 * V)
        timed('da_label_warp_dice_bwd %s' % dn, lambda: call('da_label_warp_dice_bwd', ptr(lab_m), 1, ptr(lab_t), 1, ptr(u), ptr(coef), ptr(gl), ptr(d_disp),
                                                            1, D, H, W, C, st), 26 * V)
        timed('da_warp_adjoint_labels %s' % dn, lambda: call('da_warp_adjoint_labels', ptr(lab_t), 1, ptr(u), None, ptr(B), 1, D, H, W, C, st), (C * 4 + 13) * V)
        timed('da_seg_anat_dlogits %s' % dn, lambda: call('da_seg_anat_dlogits', ptr(prob), None, 0, None, ptr(B), ptr(dlog), None, ptr(coef_a), None, ptr(gl),
                                                         1, V, C, st), (3 * C * 4) * V)
    print('loss', float(loss))


if __name__ == '__main__':
    main()
