#!/usr/bin/env python
"""Register / spill / LDS summary per kernel of a hipcc -S --cuda-device-only assembly file.  usage: python tools/kernel_regs.py file.s [substring]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for b in s.split('- .agpr_count:')[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, b).group(1)
    name = g('name')
    if flt in name or int(g('vgpr_spill_count')) > 0:
        print('%-120s vgpr %s spill %s sgpr %s' % (name[:120], g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_count')))
