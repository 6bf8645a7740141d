"""(debug helper) concat weight gradient in bf16 matrix mode: fp32-storage entry, bf16 twin, torch reference on bf16-rounded operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from deepatlas_amd import ops, _native as nat
from deepatlas_amd._native import call, ptr, stream, workspace
dev = torch.device('cuda:0')
ops.set_matrix_precision('bf16')
for (C1, C2, Cout, N, D, H, W) in [(32, 16, 16, 2, 32, 32, 32), (64, 64, 64, 2, 8, 8, 8), (64, 32, 32, 2, 16, 16, 16), (16, 0, 16, 2, 32, 32, 32)]:
    g = torch.Generator().manual_seed(1)
    rb = lambda t: t.bfloat16().float()
    x1 = rb(torch.randn((N, D, H, W, C1), generator=g)); x2 = rb(torch.randn((N, D, H, W, C2), generator=g)) if C2 else None
    dy = rb(torch.randn((N, D, H, W, Cout), generator=g))
    xcat = torch.cat([t for t in (x1, x2) if t is not None], -1).permute(0, 4, 1, 2, 3).double()
    wref = torch.nn.grad.conv3d_weight(xcat, (Cout, C1 + C2, 3, 3, 3), dy.permute(0, 4, 1, 2, 3).double(), padding=1)   # [Cout][Cin][3,3,3]
    ref = wref.permute(2, 3, 4, 1, 0).reshape(27, C1 + C2, Cout).float()
    wsb = nat.lib().da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, 1)
    wp, wn = workspace.get(wsb, dev)
    dw = torch.empty((27, C1 + C2, Cout), device=dev)
    a1, a2, gy = x1.to(dev), (x2.to(dev) if C2 else None), dy.to(dev)
    call('da_conv3d_k3_wgrad', ptr(a1), C1, ptr(a2), C2, ptr(gy), ptr(dw), None, N, D, H, W, Cout, 1, wp, wn, stream())
    e32 = float((dw.cpu() - ref).norm() / ref.norm())
    dwh = torch.empty_like(dw)
    h1, h2, hy = a1.bfloat16(), (a2.bfloat16() if C2 else None), gy.bfloat16()
    call('da_conv3d_k3_wgrad_bf16', ptr(h1), C1, ptr(h2), C2, ptr(hy), ptr(dwh), None, N, D, H, W, Cout, 1, wp, wn, stream(), 7)
    e16 = float((dwh.cpu() - ref).norm() / ref.norm())
    print((C1, C2, Cout, N, D, H, W), 'fp32-storage entry rel err %.2e   bf16 twin %.2e' % (e32, e16))
