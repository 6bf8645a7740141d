#!/usr/bin/env python
"""Fold the two passes of tools/pmc_step.sh into a per-kernel table of fabric bytes per training STEP.
Whole steps only: the dispatches between the first and the last optimiser kernel (adam_kernel) of the process, divided by the number of steps in
between (a joint step has two optimiser launches: --adam-per-step 2).
usage: python tools/pmc_step_summary.py gpurun_out/pmc_step seg [--adam-per-step N] [--out profiles/r06_step_traffic_seg]"""
import collections, csv, json, re, sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:100]


def load(path):
    """[(dispatch id, kernel, {counter: value})] in dispatch order"""
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = int(r['Dispatch_Id'])
        e = rows.setdefault(d, [short(r['Kernel_Name']), {}])
        e[1][r['Counter_Name']] = e[1].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return [(d, k, c) for d, (k, c) in sorted(rows.items())]


def whole_steps(rows, per_step):
    ad = [i for i, (_, k, _) in enumerate(rows) if k.startswith('adam_kernel')]
    if len(ad) < per_step + 1:
        raise SystemExit('fewer than two optimiser launches in the trace')
    first, last = ad[0], ad[-1]
    nsteps = (len(ad) - 1) // per_step
    last = ad[nsteps * per_step]
    return rows[first + 1:last + 1], nsteps


def main():
    src, wl = sys.argv[1], sys.argv[2]
    per_step = int(sys.argv[sys.argv.index('--adam-per-step') + 1]) if '--adam-per-step' in sys.argv else (2 if wl == 'joint' else 1)
    out = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else None
    rd, n1 = whole_steps(load('%s/%s_rd.csv' % (src, wl)), per_step)
    wr, n2 = whole_steps(load('%s/%s_wr.csv' % (src, wl)), per_step)
    tab = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for _, k, c in rd:
        tab[k][0] += 32.0 * (c.get('TCC_EA0_RDREQ_DRAM_32B_sum', 0.0) + c.get('TCC_EA0_RDREQ_GMI_32B_sum', 0.0) + c.get('TCC_EA0_RDREQ_IO_32B_sum', 0.0)) / n1
        tab[k][2] += 1
    for _, k, c in wr:
        tab[k][1] += 1024.0 * c.get('WRITE_SIZE', 0.0) / n2
    tot_r = sum(v[0] for v in tab.values()); tot_w = sum(v[1] for v in tab.values())
    alg = {'seg': 37.8e9, 'reg': 5.9e9, 'joint': 31.9e9}.get(wl)       # algorithmic bytes per step (seg: batch 2)
    lines = ['# fabric-side bytes per %s step (tools/pmc_step.sh: %d / %d whole steps in the read / write pass), kernels serialised by the counter collection' % (wl, n1, n2),
             '# reads = 32 B x (TCC_EA0_RDREQ_DRAM_32B + _GMI_32B + _IO_32B), writes = WRITE_SIZE KiB; algorithmic bytes per step (SURVEY 8d): %.1f GB' % ((alg or 0) / 1e9),
             '%-92s %8s %10s %10s %10s %7s' % ('kernel', 'launches', 'read_MB', 'write_MB', 'total_MB', 'share')]
    for k, v in sorted(tab.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
        lines.append('%-92s %8.1f %10.1f %10.1f %10.1f %6.1f%%' % (k[:92], v[2] / float(n1), v[0] / 1e6, v[1] / 1e6, (v[0] + v[1]) / 1e6, 100.0 * (v[0] + v[1]) / (tot_r + tot_w)))
    lines.append('%-92s %8s %10.1f %10.1f %10.1f' % ('TOTAL', '', tot_r / 1e6, tot_w / 1e6, (tot_r + tot_w) / 1e6))
    if alg:
        lines.append('step traffic over algorithmic: %.3f' % ((tot_r + tot_w) / alg))
    txt = '\n'.join(lines)
    print(txt)
    if out:
        open(out + '.txt', 'w').write(txt + '\n')
        json.dump({'workload': wl, 'unit': 'bytes per step', 'read': tot_r, 'write': tot_w, 'total': tot_r + tot_w, 'algorithmic': alg,
                   'over_algorithmic': (tot_r + tot_w) / alg if alg else None, 'steps_in_passes': [n1, n2],
                   'per_kernel': {k: {'read': v[0], 'write': v[1], 'launches_per_step': v[2] / float(n1)} for k, v in tab.items()}}, open(out + '.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
