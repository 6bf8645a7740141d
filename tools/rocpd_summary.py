#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share, optionally split by grid.
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--by-grid] [--top N] > profiles/rNN_xxx.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = '--by-grid' in sys.argv
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 40
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
    rows = cur.execute("select name, start, end%s from kernels" % ((', %s, %s, %s' % (gx, gx.replace('x', 'y'), gx.replace('x', 'z'))) if gx else '')).fetchall()
    agg = {}
    total = 0
    for r in rows:
        key = short(r[0]) + (('  grid=%s' % (r[3:],)) if (by_grid and gx) else '')
        d = r[2] - r[1]
        a = agg.setdefault(key, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        total += d
    print('# %s: %d dispatches, %.3f ms of kernel time' % (sys.argv[1], len(rows), total / 1e6))
    print('%-100s %8s %12s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'pct'))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%-100s %8d %12.3f %12.2f %12.2f %12.2f %6.2f%%' % (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))


if __name__ == '__main__':
    main()
