"""debug: conditioning of the UNet_light first-step gradient test -- error of the first layer's gradient norm vs the fp64 golden, both matrix
modes, under 1e-7 relative input perturbations"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from deepatlas_amd import ops
from deepatlas_amd.lib.loss import get_loss_function
from oracle import nets
import test_gpu_nets as tn
from conftest import summary_of
g = dict(np.load(os.path.join(R, 'tests', 'golden', 'seg_light.npz')))
dev = torch.device('cuda:0')
x0 = nets.closed_form_volume((1, 1, 16, 24, 32), seed=2).to(dev)
y = nets.closed_form_labels((1, 16, 24, 32), 32, seed=3).to(dev)
names = ['encoders.0.0.conv.weight', 'encoders.1.0.conv.weight', 'encoders.3.1.conv.weight', 'decoders.decBlock2.1.conv.weight']
for mode in ('fp32', 'fp32_split'):
    ops.set_matrix_precision(mode)
    for trial in range(6):
        model, sd, spec = tn._seg_model('UNET_LIGHT', 32)
        model.train(); model.lazy_head = False
        gen = torch.Generator(device='cpu').manual_seed(trial)
        x = x0 if trial == 0 else x0 * (1 + 1e-7 * torch.randn(x0.shape, generator=gen).to(dev))
        logits = model(x)
        loss = get_loss_function('dice')(n_class=32, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)(logits, y.long())
        loss.backward()
        gr = dict(model.named_parameters())
        errs = []
        for n in names:
            if f'seg_light_f64/grad/{n}' not in g: continue
            r = g[f'seg_light_f64/grad/{n}']
            errs.append('%s %.4f' % (n.split('.conv')[0][-14:], abs(summary_of(gr[n].grad)[2] - r[2]) / r[2]))
        print(mode, 'trial', trial, 'loss %.7f' % loss.item(), ' | '.join(errs))
ops.set_matrix_precision('fp32')
