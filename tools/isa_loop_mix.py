#!/usr/bin/env python
"""Instruction mix of the loops of one kernel in a hipcc -S --cuda-device-only file (static counts; loops that contain >= 10 MFMAs).
usage: python tools/isa_loop_mix.py file.s <mangled-kernel-name-substring> [--dump a b]"""
import re, sys, collections
s = open(sys.argv[1]).read()
sub = sys.argv[2]
m = re.search(r'^(\S*%s\S*):' % re.escape(sub), s, re.M)
name = m.group(1)
i = m.start(); j = s.index('s_endpgm', i)
body = s[i:j].split('\n')
print(name, len(body), 'lines')
if len(sys.argv) > 3 and sys.argv[3] == '--dump':
    a, b = int(sys.argv[4]), int(sys.argv[5])
    print('\n'.join(body[a:b])); sys.exit(0)
labels = {}
for k, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm: labels[mm.group(1)] = k
loops = []
for k, l in enumerate(body):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < k: loops.append((labels[mm.group(1)], k))
def cat(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('buffer') or op.startswith('global') or op.startswith('flat'): return 'vmem'
    return 'other'
for a, b in loops:
    seg = [x.strip() for x in body[a:b]]
    seg = [x for x in seg if x and not x.startswith(';') and not x.startswith('.')]
    n = sum('v_mfma' in x for x in seg)
    if n < 10: continue
    c = collections.Counter(x.split()[0] for x in seg)
    cc = collections.Counter()
    for op, v in c.items(): cc[cat(op)] += v
    print('loop lines %d..%d: %d instructions, %s' % (a, b, len(seg), dict(cc)))
    print('  ', ', '.join('%s %d' % kv for kv in c.most_common(60)))
