"""How often does ONE run of tests/test_gpu_nets.py::test_seg_light_first_step_vs_golden hold the gradient bound (3 x max(parameter floor, median floor))
that the split-mode re-run applies to the MEDIAN of five perturbed draws?  24 draws (the unperturbed input + 23 inputs with 1e-7 relative noise) per
matrix mode; prints the pass count and the worst ratio error / bound per mode.  python tools/debug/split_single_run.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import conftest  # noqa: F401
from oracle import nets
from deepatlas_amd import ops
from deepatlas_amd.lib.loss import get_loss_function
import test_gpu_nets as tn

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'seg_light.npz'))
dev = torch.device('cuda:0')
x0 = nets.closed_form_volume((1, 1, 16, 24, 32), seed=2).to(dev)
y = nets.closed_form_labels((1, 16, 24, 32), 32, seed=3).to(dev)


def run(x):
    model, sd, spec = tn._seg_model('UNET_LIGHT', 32)
    model.train(); model.lazy_head = True
    loss = get_loss_function('dice')(n_class=32, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)(model(x), y.long())
    loss.backward()
    return model


for mode in ('fp32_split', 'fp32'):
    prev = ops.set_matrix_precision(mode)
    model = run(x0)
    names = [n for n, _ in model.named_parameters() if not (n.endswith('conv.bias') and not n.endswith('decBlock2.2.bias'))]
    floors = {n: abs(g[f'seg_light/grad/{n}'][2] - g[f'seg_light_f64/grad/{n}'][2]) / g[f'seg_light_f64/grad/{n}'][2] for n in names}
    med = float(np.median(list(floors.values())))
    sfloor = {n: tn.rel_l2(g[f'seg_light/grad/{n}'][5:], g[f'seg_light_f64/grad/{n}'][5:]) for n in names}
    smed = float(np.median(list(sfloor.values())))
    ok, worst = 0, []
    for t in range(24):
        x = x0 if t == 0 else x0 * (1 + 1e-7 * torch.randn(x0.shape, generator=torch.Generator().manual_seed(100 + t)).to(dev))
        m = run(x)
        grads = dict(m.named_parameters())
        r = 0.0
        for n in names:
            ref64 = g[f'seg_light_f64/grad/{n}']
            e = abs(tn.summary_of(grads[n].grad)[2] - ref64[2]) / ref64[2]
            se = tn.rel_l2(tn.summary_of(grads[n].grad)[5:], ref64[5:])
            r = max(r, e / max(3 * max(floors[n], med), 1e-4), se / max(3 * max(sfloor[n], smed), 1e-4))
        ok += r < 1.0
        worst.append(r)
    print('%s: %d of 24 single runs inside the bound; error / bound: unperturbed %.2f, median %.2f, max %.2f' % (mode, ok, worst[0], float(np.median(worst)), max(worst)))
    ops.set_matrix_precision(prev)
