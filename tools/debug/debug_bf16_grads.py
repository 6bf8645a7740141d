"""(debug helper, not collected by pytest) per-parameter gradient difference between the bf16 twins and the conversion route."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from deepatlas_amd import ops
import test_gpu_bf16_storage as T
ops.set_matrix_precision('bf16'); ops.set_activation_storage('bf16')
ops.LAZY_BN = os.environ.get('DA_LAZY_BN', '1') != '0'
fused = os.environ.get('FUSED', '1') == '1'
res = []
for force in (False, True):
    ops.BF16_FORCE_BRIDGE = force
    model, sd, spec, x, y = T._seg_setup('UNET_LIGHT', 32, (32, 32, 32))
    res.append(T._device_seg_step(model, x, y, 32, fused_head=fused))
(l0, _, g0), (l1, _, g1) = res
print('loss', l0, l1)
for n in g0:
    e = float((g0[n] - g1[n]).norm() / g1[n].norm().clamp_min(1e-30))
    if e > 1e-6:
        print('%-40s %.3e  |g| %.3e' % (n, e, float(g1[n].norm())))
