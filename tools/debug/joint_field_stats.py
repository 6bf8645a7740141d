"""What the displacement field of bench.py's joint step looks like after k steps (the adjoint label scatter's cost depends on it):
per-axis spread in voxels, and the share of trilinear taps an LDS box of margin M would catch when the box is anchored at the target box
(M) or at the displaced position of its first voxel (A).  Usage: python tools/debug/joint_field_stats.py [--steps 13]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=13)
    a = ap.parse_args()
    args = argparse.Namespace(shape=[160, 192, 160], batch=2, net='UNet_light', no_fused_head=False, graph=False, precision='fp32_split')
    from deepatlas_amd import ops
    bench.set_precision(ops, 'fp32_split')
    dev = torch.device('cuda:0')
    wl, _ = bench.make_workloads(args, dev, 0, ['joint'])
    w = wl['joint']
    for k in range(a.steps):
        w.step()
    torch.cuda.synchronize()
    # the field of the next step: run the registration net forward on the same pair
    import gc
    jstep = [o for o in gc.get_objects() if type(o).__name__ == 'DeepAtlasJointStep'][0]
    x = [o for o in gc.get_objects() if torch.is_tensor(o) and o.is_cuda and o.dim() == 5 and o.shape[1] == 1 and o.shape[2:] == (160, 192, 160)]
    im = sorted(x, key=lambda t: -t.shape[0])
    im_m, im_t = im[0][:1], [t for t in im if t.shape[0] == 1][0]
    with torch.no_grad():
        out = jstep.reg(im_m, im_t)
    disp = out[1] if isinstance(out, (tuple, list)) else out
    if disp.shape[1] != 3:
        disp = [t for t in out if torch.is_tensor(t) and t.shape[1] == 3][0]
    D, H, W = disp.shape[2:]
    vox = torch.stack([disp[0, 0] * (W - 1) / 2, disp[0, 1] * (H - 1) / 2, disp[0, 2] * (D - 1) / 2])      # x, y, z displacement in voxels
    print('after %d steps: displacement in voxels  mean %s  std %s  max |.| %s' % (a.steps, [round(float(v.mean()), 3) for v in vox], [round(float(v.std()), 3) for v in vox],
                                                                                  [round(float(v.abs().max()), 2) for v in vox]))
    dx = (vox[:, :, :, 1:] - vox[:, :, :, :-1]).abs()
    print('neighbour (x) difference: mean %s  99 %% %s' % ([round(float(v.mean()), 3) for v in dx], [round(float(v.flatten()[::97].quantile(0.99)), 3) for v in dx]))
    BX, BY, BZ = 32, 8, 4
    v = vox[:, :D // BZ * BZ, :H // BY * BY, :W // BX * BX].reshape(3, D // BZ, BZ, H // BY, BY, W // BX, BX)
    anchor = v[:, :, :1, :, :1, :, :1].floor()
    for M in (1, 2, 3, 4, 6):
        inb = ((v.floor() >= -M) & (v.floor() + 1 <= M)).all(0).float().mean()
        rel = v - anchor
        ina = ((rel.floor() >= -M) & (rel.floor() + 1 <= M)).all(0).float().mean()
        print('margin %d: taps inside the grown box  anchored at the target box %.3f   at the first voxel\'s displacement %.3f' % (M, float(inb), float(ina)))


if __name__ == '__main__':
    main()
