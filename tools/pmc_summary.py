#!/usr/bin/env python
"""Fold the two rocprofv3 --pmc passes of tools/pmc_conv.sh (FETCH_SIZE, WRITE_SIZE; one counter per pass as
MI355X_MICROARCH.md prescribes) into profiles/rNN_pmc_traffic.json: HBM-side bytes per launch of each conv kernel of the
dominant layer, keyed by the C-ABI call signature bench.py reports as `roofline.kernel`.

Units / corrections: rocprofv3 reports both counters in KiB.  WRITE_SIZE of the forward kernel equals the algorithmic output
(N*V*Cout*4 B = 614400 KiB) exactly, so no write correction.  The gfx950 "FETCH_SIZE reports half" artefact applies to
128-byte requests; these kernels stage 64-byte runs (one voxel's 16-channel chunk per 4 lanes), and doubling would exceed the
bytes the kernel requests from L2 in total, so FETCH_SIZE is taken as is.
Usage: python tools/pmc_summary.py gpurun_out/pmc profiles r01"""
import collections
import csv
import json
import os
import shutil
import sys


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    layer = 'conv3d_48to16'
    per = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = os.path.join(src, '%s_%s.csv' % (layer, c))
        out = os.path.join(dst, '%s_%s_pmc_%s.csv' % (tag, layer, c.lower()))
        shutil.copyfile(f, out)
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            agg[k].append(float(r['Counter_Value']) * 1024.0)
        for k, v in agg.items():
            v = v[1:] if len(v) > 1 else v          # first launch includes cold allocation effects
            per.setdefault(k, {})[c] = sum(v) / len(v)
    sig = {'conv3_mfma_fwd_kernel<16, 1, false, false>': 'da_conv3d_k3_fwd[32, 16, 2, 160, 192, 160, 16, 1]',
           'conv3_mfma_fwd_kernel<16, 1, false, true>': 'da_conv3d_k3_fwd_bnstats[32, 16, 2, 160, 192, 160, 16, 1]',
           'conv3_mfma_fwd_kernel<16, 3, false, false>': 'da_conv3d_k3_dgrad[32, 16, 2, 160, 192, 160, 16, 1]',
           'conv3_mfma_wgrad_kernel<16, 1, false, false>': 'da_conv3d_k3_wgrad[32, 16, 2, 160, 192, 160, 16, 1]'}
    vox = 2 * 160 * 192 * 160
    alg = {'fwd': vox * (48 + 16) * 4, 'dgrad': vox * (16 + 48) * 4, 'wgrad': vox * (48 + 16) * 4 + 27 * 48 * 16 * 4}
    res = {'unit': 'bytes per launch', 'counters': 'FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes, KiB -> bytes)',
           'layer': '3x3x3 conv 48(=32+16 concat) -> 16, batch 2, 160x192x160 fp32', 'calls': {}}
    for k, name in sig.items():
        kk = k[:-1] + ', false, false>'
        if kk in per:
            per[k] = per[kk]
        if k not in per:
            continue
        f, w = per[k].get('FETCH_SIZE', 0.0), per[k].get('WRITE_SIZE', 0.0)
        a = alg[name.split('_')[3].split('[')[0]]            # 'fwd' also for da_conv3d_k3_fwd_bnstats
        res['calls'][name] = {'kernel': k, 'fetch_bytes': f, 'write_bytes': w, 'traffic_bytes': f + w,
                              'algorithmic_bytes': a, 'traffic_over_algorithmic': (f + w) / a}
    # third pass (SQ block): matrix-pipe occupancy.  SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles x (wave-level v_mfma_f32_16x16x4_f32 count),
    # summed over the 1024 SIMDs; divided by SIMDs and kernel duration it is the rate at which a SIMD's matrix pipe is busy, to be
    # read against the shader clock (2.4 GHz peak; ~1.95-2.0 GHz sustained under this load, DA_CLK probe in DESIGN.md 4.1).
    fsq = os.path.join(src, '%s_SQ.csv' % layer)
    if os.path.isfile(fsq):
        shutil.copyfile(fsq, os.path.join(dst, '%s_%s_pmc_sq_counters.csv' % (tag, layer)))
        sq = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fsq)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            sq[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
                sq[k]['duration_ns'].append(float(int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
        for k, name in sig.items():
            kk = k[:-1] + ', false, false>'                  # template arguments added since the map above was written (bf16, prologue)
            v = sq.get(kk) or sq.get(k)
            if not v or name not in res['calls']:
                continue
            m = {c: (sum(x[1:]) / len(x[1:]) if len(x) > 1 else x[0]) for c, x in v.items()}
            busy_per_simd_ghz = m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / m['duration_ns']
            res['calls'][name]['sq'] = {'mfma_busy_cycles': m['SQ_VALU_MFMA_BUSY_CYCLES'], 'duration_ms_under_pmc': m['duration_ns'] / 1e6,
                                        'mfma_busy_ghz_per_simd': busy_per_simd_ghz, 'mfma_util_vs_2p4ghz_peak': busy_per_simd_ghz / 2.4,
                                        'lds_bank_conflict_cycles': m.get('SQ_LDS_BANK_CONFLICT'), 'lds_active_cycles': m.get('SQ_LDS_IDX_ACTIVE'),
                                        'wave_cycles': m.get('SQ_WAVE_CYCLES'), 'wait_inst_any': m.get('SQ_WAIT_INST_ANY'), 'wait_any': m.get('SQ_WAIT_ANY')}
    with open(os.path.join(dst, '%s_pmc_traffic.json' % tag), 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
