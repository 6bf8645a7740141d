#!/usr/bin/env python
"""Streaming (HBM-bound) kernels on the largest BatchNorm'd tensor of the seg step (batch 2 x 160x192x160 x 16 channels, 629 MB):
bn_act_fwd (1 read + 1 write), the BN-backward reduction (2 reads) and apply (2 reads + 1 write), max-pool, against 8 TB/s.
Usage: python tools/bench_stream.py [--C 16] [--iters 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepatlas_amd import _native as nat, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--C', type=int, default=16)
    ap.add_argument('--iters', type=int, default=10)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    N, D, H, W, C = 2, 160, 192, 160, a.C
    x = torch.randn((N, C, D, H, W), device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    gamma = torch.ones(C, device=dev, requires_grad=True); beta = torch.zeros(C, device=dev, requires_grad=True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbytes = x.numel() * 4

    def once():
        y = ops.BNActFn.apply(x, gamma, beta, rm, rv, True, 0.1, 1e-5, 0.01)
        y.backward(torch.ones_like(y))
        p = ops.MaxPool2Fn.apply(x.detach().requires_grad_(True))
    once(); torch.cuda.synchronize()
    prof = nat.CallProfiler()
    nat.profiler = prof
    for _ in range(a.iters):
        once()
    torch.cuda.synchronize()
    nat.profiler = None
    passes = {'da_bn_train_stats': 1, 'da_bn_act_fwd': 2, 'da_bn_act_bwd': 5, 'da_bn_act_bwd_dbias': 5, 'da_maxpool2_fwd': 1.125}
    for (name, args), (n, ms) in sorted(prof.summary().items()):
        if name in passes:
            t = ms / n
            gb = passes[name] * nbytes / 1e9
            print('%-22s %8.3f ms  %6.2f GB  %7.1f GB/s  %.3f of HBM peak' % (name, t, gb, gb / (t * 1e-3), gb / (t * 1e-3) / 8000.0))


if __name__ == '__main__':
    main()
