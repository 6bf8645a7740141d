"""LNCC forward + backward alone at a volume size (default 160 x 192 x 160), for rocprofv3 --kernel-trace --stats runs and event timing.
Usage: python tools/bench_lncc.py [--shape D H W] [--iters 20] [--filter 9]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', type=int, nargs=3, default=[160, 192, 160])
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--filter', type=int, default=9)
    ap.add_argument('--batch', type=int, default=1)
    a = ap.parse_args()
    from deepatlas_amd.lib.loss import VoxelMorphLNCC
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    I = torch.rand((a.batch, 1) + tuple(a.shape), generator=g).to(dev).requires_grad_(True)
    J = torch.rand((a.batch, 1) + tuple(a.shape), generator=g).to(dev).requires_grad_(True)
    crit = VoxelMorphLNCC(filter_size=a.filter).to(dev)
    for _ in range(3):
        crit(I, J).backward()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(a.iters):
        e[0].record(); l = crit(I, J); e[1].record(); l.backward(); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    V = a.batch * a.shape[0] * a.shape[1] * a.shape[2]
    print('lncc F=%d %s x%d: fwd %.3f ms (%.0f GB/s algorithmic), bwd %.3f ms (%.0f GB/s)  [host-side events, include launches and autograd]'
          % (a.filter, 'x'.join(map(str, a.shape)), a.batch, tf / a.iters, 8 * V / (tf / a.iters) / 1e6, tb / a.iters, 16 * V / (tb / a.iters) / 1e6))


if __name__ == '__main__':
    main()
