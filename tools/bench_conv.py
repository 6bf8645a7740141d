#!/usr/bin/env python
"""Micro-benchmark one 3x3x3 conv layer through the C ABI (fwd / dgrad / wgrad), for rocprofv3 --pmc runs.
Usage: python tools/bench_conv.py --layer C1,C2,Cout,N,D,H,W [--what fwd,dgrad,wgrad] [--iters 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepatlas_amd import _native as nat
from deepatlas_amd._native import call, ptr, stream, workspace


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layer', default='32,16,16,2,160,192,160')
    ap.add_argument('--what', default='fwd,fwdstats,dgrad,wgrad')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--bf16-storage', action='store_true', help='bf16 activation storage: the _bf16 twins on bf16 tensors (needs DA_MATRIX_MODE=1)')
    a = ap.parse_args()
    C1, C2, Cout, N, D, H, W = [int(v) for v in a.layer.split(',')]
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    x1 = (torch.rand((N, D, H, W, C1), generator=g) * 2 - 1).to(dev)
    x2 = (torch.rand((N, D, H, W, C2), generator=g) * 2 - 1).to(dev) if C2 else None
    w = (torch.rand((27, C1 + C2, Cout), generator=g) * 0.2 - 0.1).to(dev)
    dy = (torch.rand((N, D, H, W, Cout), generator=g) * 2 - 1).to(dev)
    if os.environ.get('DA_ZERO') == '1':          # power experiment: same instruction stream on all-zero operands (no toggling in the multipliers)
        x1.zero_(); w.zero_(); dy.zero_()
        if C2: x2.zero_()
    sfx, extra = '', ()
    if a.bf16_storage:
        x1 = x1.bfloat16(); x2 = x2.bfloat16() if C2 else None; dy = dy.bfloat16()
        sfx, extra = '_bf16', (7,)
    out = torch.empty_like(dy)
    dx1 = torch.empty_like(x1); dx2 = torch.empty_like(x2) if C2 else None
    dw = torch.empty_like(w)
    pbuf = torch.empty((512, 2, Cout), dtype=torch.float64, device=dev)
    sc1, sh1 = torch.rand(C1, device=dev) + 0.5, torch.rand(C1, device=dev) - 0.5
    sc2, sh2 = (torch.rand(C2, device=dev) + 0.5, torch.rand(C2, device=dev) - 0.5) if C2 else (None, None)
    wsb = nat.lib().da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, 1)
    wp, wn = workspace.get(wsb, dev)
    st = stream()
    flops = 2.0 * 27 * (C1 + C2) * Cout * N * D * H * W
    for what in a.what.split(','):
        def run():
            if what == 'fwd':
                call('da_conv3d_k3_fwd' + sfx, ptr(x1), C1, ptr(x2), C2, ptr(w), None, ptr(out), N, D, H, W, Cout, 1, -1.0, wp, wn, st, *extra)
            elif what == 'fwdstats':
                import ctypes
                npar = ctypes.c_int(0)
                call('da_conv3d_k3_fwd_bnstats' + sfx, ptr(x1), C1, ptr(x2), C2, ptr(w), None, ptr(out), N, D, H, W, Cout, 1,
                     ptr(pbuf), 512, ctypes.byref(npar), wp, wn, st, *extra)
            elif what in ('fwdpro', 'fwdpro12'):          # input prologue on in1 (and in2): deferred BatchNorm + LeakyReLU
                import ctypes
                npar = ctypes.c_int(0)
                both = what == 'fwdpro12' and C2
                call('da_conv3d_k3_fwd_pro' + sfx, ptr(x1), C1, ptr(sc1), ptr(sh1), 0.01, ptr(x2), C2, ptr(sc2) if both else None, ptr(sh2) if both else None, 0.01,
                     ptr(w), None, ptr(out), N, D, H, W, Cout, -1.0, ptr(pbuf), 512, ctypes.byref(npar), wp, wn, st, *extra)
            elif what == 'wgradpro':
                call('da_conv3d_k3_wgrad_pro' + sfx, ptr(x1), C1, ptr(sc1), ptr(sh1), 0.01, ptr(x2), C2, None, None, -1.0, ptr(dy), ptr(dw), N, D, H, W, Cout, wp, wn, st, *extra)
            elif what == 'dgrad':
                call('da_conv3d_k3_dgrad' + sfx, ptr(dy), ptr(w), ptr(dx1), C1, ptr(dx2), C2, N, D, H, W, Cout, 1, wp, wn, st, *extra)
            else:
                call('da_conv3d_k3_wgrad' + sfx, ptr(x1), C1, ptr(x2), C2, ptr(dy), ptr(dw), None, N, D, H, W, Cout, 1, wp, wn, st, *extra)
        for _ in range(max(3, a.iters)):       # warm-up: the first launches of a process run at ramping clocks / cold TLBs (10 % slow)
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print('%s %s: %.3f ms  %.1f TFLOP/s' % (a.layer, what, ms, flops / ms / 1e9))


if __name__ == '__main__':
    main()
