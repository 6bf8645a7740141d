"""How much does the joint step's gradient error against the fp64 twin vary from run to run (the segmentation phase's adjoint scatter uses
float atomics)?  Prints, per run, the worst err / max(floor, median floor) over the tensors, per net."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_nets as T
from deepatlas_amd.optim import FlatAdam
from deepatlas_amd.models.joint import DeepAtlasJointStep
z = np.load('/root/repo/tests/golden/joint.npz')
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
for tag, C, labelled in (('c8', 8, True), ('c32', 32, True), ('c8_unlabelled_moving', 8, False)):
    worst = []
    for run in range(12):
        spec, seg_sd, reg_sd, seg, reg, (im_m, im_t, sm, st_) = T._joint_setup(C, (16, 16, 32))
        d = T.dev()
        step = DeepAtlasJointStep(seg, FlatAdam(seg.parameters(), lr=1e-3), reg, FlatAdam(reg.parameters(), lr=1e-3), C)
        step(im_m.to(d), im_t.to(d), sm.to(d) if labelled else None, st_.to(d))
        res = []
        for net, phase in ((reg, 'reg'), (seg, 'seg')):
            fl = {}
            for n, p in net.named_parameters():
                if phase == 'seg' and (n.endswith('conv.bias') or n.endswith('deconv.bias')): continue
                if p.numel() <= 4096 or phase == 'seg':
                    fl[n] = rel(z['joint/%s/grad_%s/%s' % (tag, phase, n)], z['joint/%s_f64/grad_%s/%s' % (tag, phase, n)])
            med = float(np.median(list(fl.values())))
            w = max((rel(dict(net.named_parameters())[n].grad.cpu().numpy(), z['joint/%s_f64/grad_%s/%s' % (tag, phase, n)]) / max(f, med), n) for n, f in fl.items())
            res.append((phase, round(w[0], 2), w[1]))
        worst.append(res)
    print(tag)
    for r in worst: print('   ', r)
