"""debug: per-parameter gradient error of the HIP UNet_light step against the fp64 oracle at a given volume size"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nets, steps
from deepatlas_amd.lib.network_factory import get_network
from deepatlas_amd.lib.datasets import SyntheticSegDataset
from deepatlas_amd.lib.loss import get_loss_function
from deepatlas_amd.optim import FlatAdam
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(230)
m = get_network('UNet_light')(in_channel=1, n_classes=32, bias=True, BN=True); m.weights_init()
sd32 = {k: v.detach().clone() for k, v in m.state_dict().items()}
sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd32.items()}
ds = SyntheticSegDataset(2, (S, S, S), 32, seed=230)
o32, o64 = steps.Adam(steps.trainable(sd32)), steps.Adam(steps.trainable(sd64))
m.cuda().train()
opt = FlatAdam(m.parameters(), lr=1e-3)
crit = get_loss_function('dice')(n_class=32, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
for i in (1, 0):
    img, lab, _ = ds[i]
    l32, _, g32 = steps.seg_step(sd32, o32, img[None], lab[None], nets.UNET_LIGHT, 32)
    l64, _, g64 = steps.seg_step(sd64, o64, img[None].double(), lab[None], nets.UNET_LIGHT, 32)
    opt.zero_grad()
    loss = crit(m(img[None].cuda()), lab[None].cuda())
    loss.backward()
    print('step', i, 'hip', loss.item(), 'cpu32', l32.item(), 'cpu64', l64.item())
    rows = []
    for n, p in m.named_parameters():
        ref = g64[n]
        eh = float((p.grad.cpu().double() - ref).norm() / max(ref.norm(), 1e-30))
        ec = float((g32[n].double() - ref).norm() / max(ref.norm(), 1e-30))
        rows.append((eh, ec, n, float(ref.abs().max())))
    rows.sort(reverse=True)
    for r in rows[:12]:
        print('   hip err %.3e  cpu32 err %.3e  %-40s max|g| %.2e' % r)
    opt.step()
    pd = max((float((p.detach().cpu().double() - sd64[n]).abs().max()), n) for n, p in m.named_parameters())
    print('   max param diff after step', pd)
