#!/usr/bin/env python
"""Weight gradient of a list of layer shapes through the C ABI (split mode), written to an .npz: run once with DA_WG16=1 and once with
DA_WG16=0 and compare (tools/ab/exp_w16cmp.sh) -- the eight-wave kernel against the row-owner kernel on shapes the unit tests do not reach."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from deepatlas_amd import _native as nat, ops
from deepatlas_amd._native import call, ptr, stream, workspace

SHAPES = [  # C1, C2, Cout, (N, D, H, W), lazy
    (64, 0, 8, (1, 17, 33, 47), False),
    (32, 32, 24, (1, 9, 40, 48), False),
    (16, 16, 16, (3, 5, 39, 33), True),
    (32, 16, 16, (2, 5, 24, 80), True),
    (16, 0, 12, (1, 31, 40, 16), False),
    (128, 64, 64, (1, 6, 16, 32), False),
    (64, 64, 64, (2, 8, 16, 16), True),
    (16, 0, 16, (1, 64, 64, 64), False),
]
ops.set_matrix_precision('fp32_split')
d = torch.device('cuda', 0)
out = {}
for k, (C1, C2, Cout, (N, D, H, W), lazy) in enumerate(SHAPES):
    g = torch.Generator().manual_seed(100 + k)
    x1 = (torch.randn((N, D, H, W, C1), generator=g)).to(d)
    x2 = (torch.randn((N, D, H, W, C2), generator=g)).to(d) if C2 else None
    dy = (torch.randn((N, D, H, W, Cout), generator=g)).to(d)
    dw = torch.empty((27, C1 + C2, Cout), device=d)
    wsb = nat.lib().da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, 1)
    wp, wn = workspace.get(wsb, d)
    if lazy:
        sc = (torch.rand((C1,), generator=g) * 0.5 + 0.75).to(d); sh = ((torch.rand((C1,), generator=g) - 0.5) * 0.4).to(d)
        call('da_conv3d_k3_wgrad_pro', ptr(x1), C1, ptr(sc), ptr(sh), 0.01, ptr(x2) if C2 else None, C2, None, None, -1.0, ptr(dy), ptr(dw), N, D, H, W, Cout, wp, wn, stream())
    else:
        call('da_conv3d_k3_wgrad', ptr(x1), C1, ptr(x2) if C2 else None, C2, ptr(dy), ptr(dw), None, N, D, H, W, Cout, 1, wp, wn, stream())
    torch.cuda.synchronize()
    out['s%d' % k] = dw.cpu().numpy()
np.savez(sys.argv[1], **out)
print('wrote', sys.argv[1], {k: float(np.abs(v).max()) for k, v in out.items()})
