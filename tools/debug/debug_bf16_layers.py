"""(debug helper, not collected by pytest) Per-module comparison of a UNet_light forward in bf16 storage mode: `_bf16` twins vs the conversion route (ops.BF16_FORCE_BRIDGE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepatlas_amd import ops
from oracle import nets
from deepatlas_amd.lib.network_factory import get_network

dev = torch.device('cuda:0')
ops.set_matrix_precision('bf16'); ops.set_activation_storage('bf16')
ops.LAZY_BN = os.environ.get('DA_LAZY_BN', '1') != '0'
spec = nets.UNET_LIGHT
sd = nets.closed_form_fill(nets.unet_param_shapes(1, 32, spec['encoders'], spec['decoders']), seed=1)
x = nets.closed_form_volume((2, 1, 32, 32, 32), seed=2).to(dev)
res = []
for force in (False, True):
    ops.BF16_FORCE_BRIDGE = force
    m = get_network('UNet_light')(in_channel=1, n_classes=32, bias=True, BN=True)
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    m.to(dev).train()
    seen = []
    def hook(mod, inp, out, name=None):
        t = out
        if isinstance(out, ops.LazyAct): t = out.materialize()
        if isinstance(out, tuple): t = out[0]
        if torch.is_tensor(t): seen.append((name, t.detach().float().cpu()))
    hs = [mm.register_forward_hook(lambda mod, inp, out, n=n: hook(mod, inp, out, n)) for n, mm in m.named_modules() if n and n.count('.') <= 2 and not n.endswith(('conv', 'BN', 'deconv', 'nonlinear'))]
    out = m(x)
    seen.append(('logits', out.detach().float().cpu()))
    res.append(seen)
for (n0, a), (n1, b) in zip(*res):
    e = float((a - b).norm() / b.norm().clamp_min(1e-30))
    print('%-28s %-10s rel-l2 twin vs bridge %.3e  |b| %.3e' % (n0, tuple(a.shape), e, float(b.norm())))

# the same forward through the rounding-aware oracle, block by block
rec = []
oc, od = nets._conv_bn_act, nets._deconv_bn_act
def wc(*a, **k):
    r = oc(*a, **k); rec.append((a[2], r.detach())); return r
def wd(*a, **k):
    r = od(*a, **k); rec.append((a[2], r.detach())); return r
nets._conv_bn_act, nets._deconv_bn_act = wc, wd
rnd = lambda t: t.bfloat16().float()
nets.K3_OPERAND_ROUND = rnd; nets.ACT_STORE_ROUND = rnd
sd2 = {k: v.clone() for k, v in sd.items()}
lo = nets.unet_forward(sd2, x.cpu(), spec, training=True)
rec.append(('logits', lo.detach()))
dv = dict(res[0])
for name, t in rec:
    key = name if name in dv else None
    if key is None: continue
    a = dv[key]
    print('%-28s device vs oracle rel-l2 %.3e  max|d| %.3e  |o| %.3e' % (name, float((a - t).norm() / t.norm().clamp_min(1e-30)), float((a - t).abs().max()), float(t.norm())))
