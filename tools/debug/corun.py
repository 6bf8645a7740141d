"""Do a matrix-bound convolution and an HBM-bound BatchNorm-apply pass share the GPU productively?  The 48 -> 16 forward (split mode) on one stream,
da_bn_act_fwd of a 32-channel full-resolution tensor on another: each alone, then together.  Usage: DA_MATRIX_MODE=2 python tools/debug/corun.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deepatlas_amd import _native as nat, ops
from deepatlas_amd._native import call, ptr


def main():
    dev = torch.device('cuda:0')
    ops.set_matrix_precision('fp32_split')
    N, D, H, W = 2, 160, 192, 160
    a1 = torch.rand((N, D, H, W, 32), device=dev); a2 = torch.rand((N, D, H, W, 16), device=dev)
    w = torch.rand((27, 48, 16), device=dev) - 0.5
    out = torch.empty((N, D, H, W, 16), device=dev)
    x32 = torch.rand((N, D, H, W, 32), device=dev); y32 = torch.empty_like(x32)
    sc, sf = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    wsb = nat.lib().da_conv3d_k3_ws_bytes(N, D, H, W, 48, 16, 1)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def conv(st):
        call('da_conv3d_k3_fwd', ptr(a1), 32, ptr(a2), 16, ptr(w), None, ptr(out), N, D, H, W, 16, 1, -1.0, ptr(ws), wsb, st)

    def bn(st):
        call('da_bn_act_fwd', ptr(x32), ptr(sc), ptr(sf), 0.01, ptr(y32), N * D * H * W, 32, st)

    def timed(fa, fb, iters=10):
        for _ in range(2):
            if fa: fa(sA.cuda_stream)
            if fb: fb(sB.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sA.wait_event(e0); sB.wait_event(e0)
        for _ in range(iters):
            if fa: fa(sA.cuda_stream)
            if fb: fb(sB.cuda_stream)
        torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    for rep in range(3):
        ta, tb, tab = timed(conv, None, 20), timed(None, bn, 20), timed(conv, bn, 20)
        print('conv 48->16 forward alone %.3f ms, bn_act_fwd(32 ch) alone %.3f ms, together %.3f ms (sum %.3f)' % (ta, tb, tab, ta + tb))
    # two BN passes per conv (the step has ~ 5 ms of HBM-bound passes beside ~ 15 ms of matrix kernels)
    def bn2(st):
        bn(st); bn(st)
    print('conv + 2 x bn together %.3f ms' % timed(conv, bn2, 20))


if __name__ == '__main__':
    main()
