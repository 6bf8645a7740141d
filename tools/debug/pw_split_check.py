#!/usr/bin/env python
"""Split-mode pointwise kernels (pointwise_mfma.hip, pw_split_kernel) against double: transposed conv k2 s2 forward / data gradient and the 1x1x1 conv
for several channel counts, next to the fp32 matrix instructions.  usage: python tools/debug/pw_split_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from deepatlas_amd import ops, _native as nat
from deepatlas_amd._native import call, ptr, stream, workspace

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for Cin, Cout in ((32, 32), (64, 64), (16, 32), (128, 64), (256, 128), (48, 48), (64, 16), (64, 32), (64, 48), (48, 64), (32, 64)):
    N, D, H, W = 1, 4, 6, 8
    x = (torch.rand((N, Cin, D, H, W), generator=g) * 2 - 1)
    w = (torch.rand((Cin, Cout, 2, 2, 2), generator=g) * 2 - 1) * 0.2
    gy = (torch.rand((N, Cout, 2 * D, 2 * H, 2 * W), generator=g) * 2 - 1)
    xd = x.double().requires_grad_(True)
    yd = F.conv_transpose3d(xd, w.double(), None, stride=2)
    yd.backward(gy.double())
    res = {}
    for mode in ('fp32', 'fp32_split'):
        ops.set_matrix_precision(mode)
        a = x.to(dev).permute(0, 2, 3, 4, 1).contiguous()
        w_tio = w.to(dev).permute(2, 3, 4, 0, 1).reshape(8, Cin, Cout).contiguous()
        y = torch.empty((N, 2 * D, 2 * H, 2 * W, Cout), device=dev)
        wsb = nat.lib().da_pointwise_ws_bytes(8, Cin, Cout)
        wp, wn = workspace.get(wsb, dev)
        call('da_deconv_k2s2_fwd', ptr(a), ptr(w_tio), None, ptr(y), N, D, H, W, Cin, Cout, wp, wn, stream())
        gyd = gy.to(dev).permute(0, 2, 3, 4, 1).contiguous()
        dx = torch.empty_like(a)
        call('da_deconv_k2s2_dgrad', ptr(gyd), ptr(w_tio), ptr(dx), N, D, H, W, Cin, Cout, wp, wn, stream())
        torch.cuda.synchronize()
        rel = lambda u, v: float((u.double().cpu() - v).norm() / v.norm())
        res[mode] = (rel(y.permute(0, 4, 1, 2, 3), yd.detach()), rel(dx.permute(0, 4, 1, 2, 3), xd.grad))
    ops.set_matrix_precision('fp32')
    print('Cin %3d Cout %3d  fwd rel-l2: fp32 %.2e split %.2e   dgrad: fp32 %.2e split %.2e' % (Cin, Cout, res['fp32'][0], res['fp32_split'][0], res['fp32'][1], res['fp32_split'][1]))

# which dy channel group of the K = 64 data gradient is off: gy non-zero in one 16-channel group at a time
Cin, Cout, N, D, H, W = 32, 64, 1, 4, 6, 8
w = (torch.rand((Cin, Cout, 2, 2, 2), generator=g) * 2 - 1) * 0.2
for grp in range(4):
    gy = torch.zeros((N, Cout, 2 * D, 2 * H, 2 * W))
    gy[:, 16 * grp:16 * grp + 16] = torch.rand((N, 16, 2 * D, 2 * H, 2 * W), generator=g) * 2 - 1
    xd = torch.zeros((N, Cin, D, H, W), dtype=torch.float64, requires_grad=True)
    F.conv_transpose3d(xd, w.double(), None, stride=2).backward(gy.double())
    ops.set_matrix_precision('fp32_split')
    w_tio = w.to(dev).permute(2, 3, 4, 0, 1).reshape(8, Cin, Cout).contiguous()
    gyd = gy.to(dev).permute(0, 2, 3, 4, 1).contiguous()
    dx = torch.empty((N, D, H, W, Cin), device=dev)
    wp, wn = workspace.get(nat.lib().da_pointwise_ws_bytes(8, Cin, Cout), dev)
    call('da_deconv_k2s2_dgrad', ptr(gyd), ptr(w_tio), ptr(dx), N, D, H, W, Cin, Cout, wp, wn, stream())
    torch.cuda.synchronize()
    ops.set_matrix_precision('fp32')
    print('dy group %d: rel-l2 %.2e' % (grp, float((dx.permute(0, 4, 1, 2, 3).double().cpu() - xd.grad).norm() / xd.grad.norm())))
