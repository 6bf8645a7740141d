"""debug: split mode vs fp64 on tiny volumes / UNet_light channel configs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch, torch.nn.functional as F
from deepatlas_amd import ops
from conftest import rel_l2, max_abs_rel
dev = torch.device('cuda:0')
def cl(x): return x.to(dev).contiguous(memory_format=torch.channels_last_3d)
def rnd(shape, seed):
    g = torch.Generator().manual_seed(seed); return torch.rand(shape, generator=g) * 2 - 1
for dims in [(1, 2, 3, 4), (1, 4, 6, 8), (1, 8, 12, 16), (1, 16, 24, 32), (2, 5, 9, 17)]:
    for (C1, C2, Cout) in [(8, 0, 16), (16, 0, 16), (16, 0, 32), (32, 0, 32), (32, 0, 64), (64, 0, 64), (64, 64, 64), (64, 32, 32), (32, 16, 16)]:
        N, D, H, W = dims
        x1 = rnd((N, C1, D, H, W), 1); x2 = rnd((N, C2, D, H, W), 2) if C2 else None
        w = rnd((Cout, C1 + C2, 3, 3, 3), 3) * 0.2; b = rnd((Cout,), 4) * 0.1; go = rnd((N, Cout, D, H, W), 5)
        xr1 = x1.double().requires_grad_(True); xr2 = x2.double().requires_grad_(True) if C2 else None
        wr = w.double().requires_grad_(True)
        yr = F.conv3d(torch.cat((xr1, xr2), 1) if C2 else xr1, wr, b.double(), padding=1); yr.backward(go.double())
        res = {}
        for mode in ('fp32', 'fp32_split'):
            ops.set_matrix_precision(mode)
            a1 = cl(x1).requires_grad_(True); a2 = cl(x2).requires_grad_(True) if C2 else None
            wg = w.to(dev).requires_grad_(True)
            y = ops.Conv3dK3Fn.apply(a1, a2, wg, b.to(dev), 1, -1.0); y.backward(cl(go)); torch.cuda.synchronize()
            res[mode] = (rel_l2(y.detach().cpu().double().numpy(), yr.detach().numpy()), rel_l2(a1.grad.cpu().double().numpy(), xr1.grad.numpy()),
                         rel_l2(wg.grad.cpu().double().numpy(), wr.grad.numpy()))
        ops.set_matrix_precision('fp32')
        flag = '  <<<<' if max(res['fp32_split']) > 3 * max(max(res['fp32']), 1e-7) else ''
        print(dims, (C1, C2, Cout), 'fp32 %.1e %.1e %.1e | split %.1e %.1e %.1e%s' % (res['fp32'] + res['fp32_split'] + (flag,)))
