#!/usr/bin/env python
"""How long does a tiny kernel on the main stream take while a whole-GPU persistent kernel runs on a side stream?
(found in the reg step: a 4-workgroup pack kernel took 300 us beside the flow conv's weight gradient.)
Usage: python tools/debug/concurrency_probe.py [--layer 8,16,3,1,160,192,160] [--what wgrad]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deepatlas_amd import _native as nat
from deepatlas_amd._native import call, ptr, workspace


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layer', default='8,16,3,1,160,192,160')
    ap.add_argument('--what', default='wgrad')
    a = ap.parse_args()
    C1, C2, Cout, N, D, H, W = [int(v) for v in a.layer.split(',')]
    dev = torch.device('cuda:0')
    x1 = torch.rand((N, D, H, W, C1), device=dev)
    x2 = torch.rand((N, D, H, W, C2), device=dev) if C2 else None
    w = torch.rand((27, C1 + C2, Cout), device=dev) * 0.1
    dy = torch.rand((N, D, H, W, Cout), device=dev)
    dw = torch.empty_like(w); out = torch.empty_like(dy)
    dx1 = torch.empty_like(x1); dx2 = torch.empty_like(x2) if C2 else None
    wsb = nat.lib().da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, 1)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    small = torch.zeros(1024, device=dev)
    small2 = torch.zeros(1 << 20, device=dev)
    # a small LDS-using library kernel on the main stream: the folded up-sampling conv's data gradient on a tiny volume (pack + 1-tile kernel)
    uw = torch.rand((27, 8, 8), device=dev) * 0.1
    udy = torch.rand((1, 8, 16, 32, 8), device=dev)
    udx = torch.empty((1, 4, 8, 16, 8), device=dev)
    uwsb = 1 << 22
    uws = torch.empty(uwsb, dtype=torch.uint8, device=dev)

    def big(st):
        if a.what == 'wgrad':
            call('da_conv3d_k3_wgrad', ptr(x1), C1, ptr(x2), C2, ptr(dy), ptr(dw), None, N, D, H, W, Cout, 1, ptr(ws), wsb, st)
        elif a.what == 'fwd':
            call('da_conv3d_k3_fwd', ptr(x1), C1, ptr(x2), C2, ptr(w), None, ptr(out), N, D, H, W, Cout, 1, -1.0, ptr(ws), wsb, st)
        else:
            call('da_conv3d_k3_dgrad', ptr(dy), ptr(w), ptr(dx1), C1, ptr(dx2), C2, N, D, H, W, Cout, 1, ptr(ws), wsb, st)

    def probes():
        return {
            'fill_1k': lambda: small.fill_(1.0),
            'add_1M': lambda: small2.add_(1.0),
            'up_dgrad_tiny': lambda: call('da_upconv3d_k3_dgrad', ptr(udy), ptr(uw), ptr(udx), 8, None, 0, 1, 4, 8, 16, 8, ptr(uws), uwsb, main.cuda_stream),
        }

    for _ in range(3):
        big(main.cuda_stream)
        for f in probes().values(): f()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record(); big(main.cuda_stream); e[1].record(); torch.cuda.synchronize()
    print('%s %s alone: %.3f ms' % (a.layer, a.what, e[0].elapsed_time(e[1])))
    for name, f in probes().items():
        e[0].record(); f(); e[1].record(); torch.cuda.synchronize()
        alone = e[0].elapsed_time(e[1])
        res = []
        for rep in range(3):
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
                s0.record(side); big(side.cuda_stream); s1.record(side)
            torch.cuda._sleep(200000)          # ~0.1 ms on the main stream: the probe starts while the big kernel is running
            e[2].record(); f(); e[3].record()
            torch.cuda.synchronize()
            res.append((e[2].elapsed_time(e[3]), s0.elapsed_time(s1), s0.elapsed_time(e[2])))
        print('%-14s alone %.3f ms; beside the big kernel: %s  (probe ms, big ms, probe start after big start ms)' % (name, alone, ['%.3f/%.3f/%.3f' % r for r in res]))


if __name__ == '__main__':
    main()
