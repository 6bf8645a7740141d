#!/usr/bin/env python
"""HBM-bound rows at 160x192x160 (one sample): per C-ABI call time (HIP events on the launch stream) against the
algorithmic bytes of SURVEY.md §8(d) -> achieved GB/s and fraction of the 8 TB/s HBM peak.
Usage: python tools/bench_losses.py [--iters 5]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepatlas_amd import _native as nat, ops
from deepatlas_amd.lib import loss as L, evalMetrics as em
from deepatlas_amd.lib.datasets import structured_labels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--shape', type=int, nargs=3, default=[160, 192, 160])
    a = ap.parse_args()
    D, H, W = a.shape
    V = D * H * W
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(230)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last_3d)
    img1, img2 = torch.rand((1, 1, D, H, W), generator=g).to(dev), torch.rand((1, 1, D, H, W), generator=g).to(dev)
    disp = cl((torch.rand((1, 3, D, H, W), generator=g) - 0.5) * 0.05)
    logits = cl(torch.randn((1, 32, D, H, W), generator=g))
    labels = structured_labels((D, H, W), 32).to(dev)[None]
    labels2 = structured_labels((D, H, W), 32, seed=3).to(dev)[None]
    # algorithmic bytes per call (fp32): name -> bytes
    alg = {
        'da_dice_fwd': (32 * 4 + 1) * V, 'da_dice_bwd': (2 * 32 * 4 + 1) * V,
        'da_ncc_fwd': 2 * 4 * V, 'da_ncc_bwd': 3 * 4 * V,
        'da_bending_fwd': 3 * 4 * V, 'da_bending_bwd': 6 * 4 * V,
        'da_gradloss_fwd': 3 * 4 * V, 'da_gradloss_bwd': 6 * 4 * V,
        'da_lncc_fwd': 2 * 4 * V, 'da_lncc_bwd': 4 * 4 * V,                 # read I, J (+ write dI, dJ); intermediates are implementation traffic
        'da_warp_fwd[1]': 5 * 4 * V, 'da_warp_bwd[1]': 8 * 4 * V,
        'da_warp_fwd[32]': 67 * 4 * V, 'da_warp_bwd[32]': (70 + 32) * 4 * V,
        'da_argmax_dice_counts': (32 * 4 + 1) * V, 'da_label_overlap_counts': 2 * V,
        # round 2: fused kernels (bytes = what has to cross HBM once)
        'da_head_dice_fwd': (16 * 4 + 1) * V, 'da_head_dice_bwd': (2 * 16 * 4 + 1) * V,                 # read x (+ write dx), labels
        'da_label_warp_dice_fwd': (12 + 2) * V, 'da_label_warp_dice_bwd': (12 + 2 + 12) * V,            # disp + two label maps (+ d_disp)
        'da_warp_adjoint_labels': (32 * 4 + 12 + 1) * V,                                                # B written once, disp + labels read
        'da_seg_anat_dlogits': (3 * 32 * 4 + 1) * V,                                                    # prob, B read; d logits written
        'da_xent_fwd': (32 * 4 + 1) * V, 'da_xent_bwd': (2 * 32 * 4 + 1) * V,
        'da_act_bwd_add_dbias': 4 * 16 * 4 * V,                                                         # g1, g2, y read; dx written (16 channels)
    }
    prof = nat.CallProfiler()
    crit_d = L.DiceLossMultiClass(n_class=32, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
    crit_n, crit_b, crit_g, crit_l = L.NormalizedCrossCorrelationLoss(), L.BendingEnergyLoss(), L.gradientLoss(), L.VoxelMorphLNCC().to(dev)
    src32 = cl(torch.rand((1, 32, D, H, W), generator=g))

    x16 = cl(torch.rand((1, 16, D, H, W), generator=g))
    w_head, b_head = (torch.rand((32, 16, 1, 1, 1), generator=g) - 0.5).to(dev), (torch.rand((32,), generator=g) - 0.5).to(dev)
    lab8 = labels.to(torch.uint8)
    lab8b = labels2.to(torch.uint8)
    ce = L.get_loss_function('cross_entropy')()
    g16a, g16b = cl(torch.rand((1, 16, D, H, W), generator=g)), cl(torch.rand((1, 16, D, H, W), generator=g))

    def once():
        x = logits.clone().requires_grad_(True); crit_d(x, labels.long()).backward()
        xh = x16.clone().requires_grad_(True); wh = w_head.clone().requires_grad_(True)
        ops.HeadDiceFn.apply(xh, wh, b_head, lab8, 'Uniform', False, 1e-6).backward()
        u = disp.clone().requires_grad_(True); ops.LabelWarpDiceFn.apply(lab8, lab8b, u, 32, 'Uniform', False, 1e-6).backward()
        x = logits.clone().requires_grad_(True)
        ls, la = ops.SegPhaseLossFn.apply(x, lab8, disp, lab8b, 'Uniform', False, 1e-6); (ls + la).backward()
        x = logits.clone().requires_grad_(True); ce(x, lab8).backward()
        ya, yb = ops.Conv3dK3Fn.apply(x16[:, :8], None, torch.zeros((16, 8, 3, 3, 3), device=dev, requires_grad=True), None, 1, 0.0, False, True)
        (ya * g16a + yb * g16b).sum().backward()
        i1 = img1.clone().requires_grad_(True); crit_n(i1, img2).backward()
        u = disp.clone().requires_grad_(True); crit_b(u).backward()
        u = disp.clone().requires_grad_(True); crit_g(u).backward()
        i1 = img1.clone().requires_grad_(True); i2 = img2.clone().requires_grad_(True); crit_l(i1, i2).backward()
        # the image warp's source never needs a gradient (registration step: d_disp only); the 32-channel probability warp does
        for s, need in ((img1, False), (src32, True)):
            sr = s.clone().requires_grad_(need); u = disp.clone().requires_grad_(True)
            out, _ = ops.WarpFn.apply(sr, u); out.sum().backward()
        em.eval_dice_counts(logits, labels)
        ops.label_overlap_counts(labels, labels2, 32)

    once(); torch.cuda.synchronize()
    nat.profiler = prof
    for _ in range(a.iters):
        once()
    torch.cuda.synchronize()
    nat.profiler = None
    rows = {}
    for (name, args), (n, ms) in prof.summary().items():
        key = name
        if name.startswith('da_warp_'):
            key = '%s[%d]' % (name, args[4])
        if key in alg:
            r = rows.setdefault(key, [0, 0.0]); r[0] += n; r[1] += ms
    out = []
    for k, (n, ms) in sorted(rows.items()):
        t = ms / n
        gbs = alg[k] / (t * 1e-3) / 1e9
        out.append(dict(call=k, avg_ms=round(t, 4), algorithmic_MB=round(alg[k] / 1e6, 1), GBps=round(gbs, 1), frac_of_8TBps=round(gbs / 8000.0, 3)))
        print('%-28s %8.3f ms  %9.1f MB  %8.1f GB/s  %.3f of HBM peak' % (k, t, alg[k] / 1e6, gbs, gbs / 8000.0))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
