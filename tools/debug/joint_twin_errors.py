import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_nets as T
from deepatlas_amd.optim import FlatAdam
from deepatlas_amd.models.joint import DeepAtlasJointStep
z = np.load('/root/repo/tests/golden/joint.npz')
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
for tag, C, labelled in (('c8', 8, True), ('c32', 32, True), ('c8_unlabelled_moving', 8, False)):
    spec, seg_sd, reg_sd, seg, reg, (im_m, im_t, sm, st_) = T._joint_setup(C, (16, 16, 32))
    d = T.dev()
    step = DeepAtlasJointStep(seg, FlatAdam(seg.parameters(), lr=1e-3), reg, FlatAdam(reg.parameters(), lr=1e-3), C)
    out = step(im_m.to(d), im_t.to(d), sm.to(d) if labelled else None, st_.to(d))
    print(tag, {k: (float(out[k]), float(z['joint/%s_f64/%s' % (tag, k)])) for k in ('sim', 'bend', 'anat_reg', 'sup', 'anat_seg')})
    rows = []
    for n, p in seg.named_parameters():
        g64, g32 = z['joint/%s_f64/grad_seg/%s' % (tag, n)], z['joint/%s/grad_seg/%s' % (tag, n)]
        if n.endswith('conv.bias') or n.endswith('deconv.bias'):
            continue
        rows.append((n, rel(p.grad.cpu().numpy(), g64), rel(g32, g64)))
    rows.sort(key=lambda r: -r[1] / max(r[2], 1e-12))
    print('  seg: worst dev/floor ratios:', [(n, '%.1e' % e, '%.1e' % f) for n, e, f in rows[:5]])
    print('  seg: max dev err %.2e, max floor %.2e, median floor %.2e' % (max(r[1] for r in rows), max(r[2] for r in rows), float(np.median([r[2] for r in rows]))))
    rr = []
    for n, p in reg.named_parameters():
        k = 'joint/%s_f64/grad_reg/%s' % (tag, n)
        if p.numel() <= 4096:
            rr.append((n, rel(p.grad.cpu().numpy(), z[k]), rel(z['joint/%s/grad_reg/%s' % (tag, n)], z[k])))
    rr.sort(key=lambda r: -r[1])
    print('  reg (small tensors): max dev err %.2e (%s), its floor %.2e' % (rr[0][1], rr[0][0], rr[0][2]))
