#!/usr/bin/env python
"""Timeline view of a rocprofv3 (rocpd sqlite) kernel trace of a two-stream training step: how much of the wall time has a
matrix (MFMA conv) kernel in flight, how much only HBM-bound kernels, how much nothing, and which kernels account for the time
that no matrix kernel covers ("exposed" time).
Usage: python tools/rocpd_timeline.py x_results.db [--skip-frac 0.3] [--top 25] [--dump] [--adam-per-step 2]
(--adam-per-step: optimiser launches per training step -- 1 for seg / reg, 2 for the joint step's two phases; the window is cut at optimiser launches.)"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:80]


def is_matrix(name):
    return name.startswith('conv3_mfma_') or name.startswith('conv3_fwdsp') or name.startswith('conv3_split_wgrad') or name.startswith('s2n_') or name.startswith('up_fwd') or name.startswith('up_dgrad') or name.startswith('up_wgrad_kernel')


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = float(sys.argv[sys.argv.index('--skip-frac') + 1]) if '--skip-frac' in sys.argv else 0.3
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 25
    rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t0 + (t1 - t0) * skip                      # drop warm-up at the head of the trace
    aps = int(sys.argv[sys.argv.index('--adam-per-step') + 1]) if '--adam-per-step' in sys.argv else 1
    adam = [e for n, s, e, q in rows if 'adam_kernel' in n]
    nsteps = 0
    if len(adam) >= 3 * aps:                         # steady state: from the end of an optimiser launch to the end of the last one, whole steps only
        k = max(1, len(adam) // (2 * aps)) * aps
        lo, t1, nsteps = adam[-k - 1], adam[-1], k // aps
    rows = [(short(n), max(s, lo), min(e, t1), q) for n, s, e, q in rows if e > lo and s < t1]
    ev = []                                          # sweep line over [start, end) of every kernel
    for i, (n, s, e, q) in enumerate(rows):
        ev.append((s, 1, i)); ev.append((e, 0, i))
    ev.sort()
    active = set(); nm = 0
    prev = lo
    tm = tn = ti = 0
    exposed = defaultdict(float)
    gaps = defaultdict(lambda: [0.0, 0])             # idle time by (kernel that ended before the gap -> kernel that started after it)
    last_ended = '(start)'
    for t, kind, i in ev:
        dt = t - prev
        if dt > 0:
            if nm > 0: tm += dt
            elif active:
                tn += dt
                for j in active: exposed[rows[j][0]] += dt / len(active)
            else:
                ti += dt
                if kind == 1:
                    g = gaps[(last_ended, rows[i][0])]; g[0] += dt; g[1] += 1
        prev = t
        if kind == 1:
            active.add(i); nm += is_matrix(rows[i][0])
        else:
            active.discard(i); nm -= is_matrix(rows[i][0]); last_ended = rows[i][0]
    wall = prev - lo
    print('# %s: window %.3f ms = %d steps, queues %s' % (sys.argv[1], wall / 1e6, nsteps, sorted(set(r[3] for r in rows))))
    if nsteps:
        wall_step = wall / nsteps
        print('per step: %.3f ms' % (wall_step / 1e6))
    print('matrix kernel in flight      %8.3f ms  %5.1f %%' % (tm / 1e6, 100 * tm / wall))
    print('only other kernels in flight %8.3f ms  %5.1f %%' % (tn / 1e6, 100 * tn / wall))
    print('nothing in flight            %8.3f ms  %5.1f %%' % (ti / 1e6, 100 * ti / wall))
    print('exposed time by kernel (no matrix kernel running beside it):')
    for k, v in sorted(exposed.items(), key=lambda kv: -kv[1])[:top]:
        print('  %-80s %8.3f ms  %5.1f %%' % (k, v / 1e6, 100 * v / wall))
    print('idle gaps (nothing in flight) by the kernels on either side:')
    for (a, b), (v, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
        print('  %8.3f ms  %4d x  %-44s -> %s' % (v / 1e6, c, a[:44], b[:60]))


def dump_last_step(path, min_us=150.0):      # (--dump-all: every kernel)
    """--dump: start / end / queue / duration of every kernel longer than min_us in the last optimiser step of the trace."""
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
    seg = rows[adam[-2] + 1:adam[-1] + 1]
    t0 = seg[0][1]
    for n, s_, e, q in seg:
        d = (e - s_) / 1e3
        if d > min_us:
            print('%7.3f -> %7.3f  q%d  %8.1f us  %s' % ((s_ - t0) / 1e6, (e - t0) / 1e6, q, d, short(n)[:60]))


if __name__ == '__main__':
    if '--dump' in sys.argv:
        dump_last_step(sys.argv[1], 0.0 if '--dump-all' in sys.argv else 150.0)
        sys.exit(0)
    main()
