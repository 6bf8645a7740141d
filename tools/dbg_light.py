import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from conftest import summary_of
from oracle import nets
from deepatlas_amd.lib.network_factory import get_network
from deepatlas_amd.lib.loss import get_loss_function
g = dict(np.load('tests/golden/seg_light.npz'))
spec = nets.UNET_LIGHT
sd = nets.closed_form_fill(nets.unet_param_shapes(1, 32, spec['encoders'], spec['decoders']), seed=1)
model = get_network('UNet_light')(in_channel=1, n_classes=32, bias=True, BN=True)
model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
model.cuda().train()
x = nets.closed_form_volume((1, 1, 16, 24, 32), seed=2).cuda()
y = nets.closed_form_labels((1, 16, 24, 32), 32, seed=3).cuda()
logits = model(x)
loss = get_loss_function('dice')(n_class=32, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)(logits, y.long())
loss.backward()
rows = []
for n, p in model.named_parameters():
    if n.endswith('conv.bias') and not n.endswith('decBlock2.2.bias'): continue
    ref, ref64 = g['seg_light/grad/' + n], g['seg_light_f64/grad/' + n]
    rows.append((abs(summary_of(p.grad)[2] - ref64[2]) / ref64[2], abs(ref[2] - ref64[2]) / ref64[2], n))
rows.sort(reverse=True)
for r in rows[:8]: print('%.3e (ref floor %.3e) %s' % r)
print('median err %.3e' % np.median([r[0] for r in rows]))
