#!/usr/bin/env python
"""Print the few numbers of a bench.py JSON line that matter while iterating.  usage: python tools/bench_brief.py file.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print('headline %.2f %s  %.3f ms/step  (%s)' % (d['value'], d['unit'], d['ms_per_step'], d['config'].get('matrix_precision')))
r = d.get('roofline') or {}
print('roofline', r.get('kernel'), r.get('avg_ms'), 'ms frac', r.get('frac'), 'step_frac', r.get('step_frac'))
pp = r.get('post_run_pass') or {}
for row in pp.get('roofline_layer', []):
    print('   ', row['call'], row['avg_ms'], 'ms', row['tflops'], 'TFLOP/s', row['frac'])
print('    all conv calls', pp.get('all_conv_calls'))
for k, v in (d.get('extra') or {}).items():
    if isinstance(v, dict) and 'value' in v:
        print('extra', k, v['value'], v['unit'], v['ms_per_step'], 'ms')
    elif isinstance(v, dict):
        for k2, v2 in v.items():
            if isinstance(v2, dict) and 'value' in v2: print('extra', k, k2, v2['value'], v2['unit'], v2['ms_per_step'], 'ms')
