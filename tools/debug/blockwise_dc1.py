"""debug: full-UNet block dc1 in split mode -- is the dW mismatch a ReLU kink (pre-activation ~0 taking the other sign)?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, torch.nn.functional as F
from deepatlas_amd import ops
import test_gpu_nets as tn
for mode in ('fp32', 'fp32_split'):
    ops.set_matrix_precision(mode)
    model, sd, x, y = tn._unet_full(True)
    outs, gouts = {}, {}
    def fh(name):
        def hook(m, i, o):
            outs[name] = o.detach().cpu()
            o.register_hook(lambda gr, name=name: gouts.__setitem__(name, gr.detach().cpu().contiguous()))
        return hook
    for n in ('dc2', 'dc1'):
        getattr(model, n).register_forward_hook(fh(n))
    from deepatlas_amd.lib.loss import get_loss_function
    logits = model(x)
    get_loss_function('dice')(n_class=3, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)(logits, y.long()).backward()
    P = dict(model.named_parameters())
    xin = outs['dc2'].double().requires_grad_(True)
    w, b = sd['dc1.0.weight'].double().requires_grad_(True), sd['dc1.0.bias'].double()
    ga, be = sd['dc1.1.weight'].double(), sd['dc1.1.bias'].double()
    z = F.batch_norm(F.conv_transpose3d(xin, w, b, stride=1, padding=1), None, None, ga, be, True, 0.1, 1e-5)
    yy = F.relu(z)
    dev_mask = outs['dc1'] > 0
    flips = (dev_mask != (z.detach() > 0))
    print(mode, 'shape', tuple(z.shape), 'flipped ReLU elements', int(flips.sum()), 'min |z| at flips', float(z.detach().abs()[flips].min()) if flips.any() else None,
          'smallest |z| overall %.2e' % float(z.detach().abs().min()))
    yy.backward(gouts['dc1'].double())
    rel = lambda a, c: float((a.double() - c.double()).norm() / c.double().norm())
    print('   dW rel (fp64 mask) %.3e' % rel(P['dc1.0.weight'].grad.cpu(), w.grad))
    w2 = sd['dc1.0.weight'].double().requires_grad_(True)
    z2 = F.batch_norm(F.conv_transpose3d(xin.detach(), w2, b, stride=1, padding=1), None, None, ga, be, True, 0.1, 1e-5)
    (z2 * dev_mask.double()).backward(gouts['dc1'].double())
    print('   dW rel (device mask) %.3e' % rel(P['dc1.0.weight'].grad.cpu(), w2.grad))
ops.set_matrix_precision('fp32')
