"""debug: async (side-stream) weight gradients against the synchronous path at a given size -- gradients after one backward"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepatlas_amd import ops
from deepatlas_amd.lib.network_factory import get_network
from deepatlas_amd.lib.datasets import SyntheticSegDataset
from deepatlas_amd.lib.loss import get_loss_function
from deepatlas_amd.optim import FlatAdam
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ds = SyntheticSegDataset(2, (S, S, S), 32, seed=230)
res = []
for flag in (False, True, True):
    ops.enable_async_wgrad(flag)
    torch.manual_seed(230)
    m = get_network('UNet_light')(in_channel=1, n_classes=32, bias=True, BN=True); m.weights_init()
    m.cuda().train()
    opt = FlatAdam(m.parameters(), lr=1e-3)
    crit = get_loss_function('dice')(n_class=32, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
    losses = []
    grads = None
    for i in (0, 1, 0):
        img, lab, _ = ds[i]
        opt.zero_grad()
        loss = crit(m(img[None].cuda()), lab[None].cuda())
        loss.backward()
        if grads is None:
            ops.join_side_stream(); torch.cuda.synchronize()
            grads = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        opt.step()
        losses.append(loss.item())
    res.append((losses, grads))
    print('async' if flag else 'sync ', losses)
for n in res[0][1]:
    a, b, c = res[0][1][n], res[1][1][n], res[2][1][n]
    if not torch.equal(a, b) or not torch.equal(b, c):
        print('DIFF %-40s sync-vs-async %.3e  async-vs-async %.3e  |g| %.3e' % (n, float((a - b).abs().max()), float((b - c).abs().max()), float(a.abs().max())))
