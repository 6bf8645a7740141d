"""Host-side cost of a training step: cProfile over k steps of bench.py's workload (which Python functions the launch thread spends its time in).
Usage: python tools/debug/host_profile.py [seg|reg|joint] [--steps 8]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('workload', nargs='?', default='seg')
    ap.add_argument('--steps', type=int, default=8)
    a = ap.parse_args()
    args = argparse.Namespace(shape=[160, 192, 160], batch=2, net='UNet_light', no_fused_head=False, graph=False, precision='fp32_split')
    from deepatlas_amd import ops
    bench.set_precision(ops, 'fp32_split')
    wl, _ = bench.make_workloads(args, torch.device('cuda:0'), 0, [a.workload])
    w = wl[a.workload]
    for _ in range(5):
        w.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        w.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%s: host issue %.3f ms/step, wall %.3f ms/step' % (a.workload, (t1 - t0) / a.steps * 1e3, (t2 - t0) / a.steps * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        w.step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(22)
    st.sort_stats('cumulative').print_stats(30)


if __name__ == '__main__':
    main()
