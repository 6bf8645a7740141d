#!/usr/bin/env python
"""Fold the two rocprofv3 --pmc passes of tools/pmc_conv.sh (FETCH_SIZE, WRITE_SIZE; one counter per pass as
MI355X_MICROARCH.md prescribes) into profiles/rNN_pmc_traffic.json: HBM-side bytes per launch of each conv kernel of the
dominant layer, keyed by the C-ABI call signature bench.py reports as `roofline.kernel`.

Units / corrections: rocprofv3 reports both counters in KiB.  WRITE_SIZE of the forward kernel equals the algorithmic output
(N*V*Cout*4 B = 614400 KiB) exactly, so no write correction.  FETCH_SIZE is CALIBRATED (MI355X_MICROARCH.md: gfx950 reports half the
bytes of a wide coalesced streaming read; other widths uncalibrated): tools/ubench/fetch_calib.hip streams 1 GiB once in the two request
shapes of the halo staging (contiguous 16 B / lane; 64-byte runs at a 128-byte stride) under the same counter, and the measured
requested-bytes / FETCH_SIZE factors are applied to the conv kernels' raw FETCH_SIZE in proportion to the bytes each shape stages.
Usage: python tools/pmc_summary.py gpurun_out/pmc profiles r02"""
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
            agg[k + '@' + r['Grid_Size']].append(float(r['Counter_Value']) * 1024.0)          # same kernel, different launch shape (split mode: forward vs data gradient)
        for k, v in agg.items():
            v = v[1:] if len(v) > 1 else v          # first launch includes cold allocation effects
            per.setdefault(k, {})[c] = sum(v) / len(v)
    # kernel template instantiations at HEAD: fwd <CK, NREP, MASKED, STATS, BF, PRO, DYN, SP, S2F, PAIR, HB>, wgrad <CK, NREP, YS, MASKED, BF, PRO, SP, HB>,
    # split weight gradient conv3_split_wgrad_kernel<PRO, NPL, HB>
    mode = 0
    try:
        mode = int(open(os.path.join(src, 'matrix_mode.txt')).read().strip())
    except (OSError, ValueError):
        pass
    if mode == 2:
        # (the data gradient 16 -> 48 runs the forward kernel with one N-tile per workgroup and three cout groups: 168 x 3 workgroups)
        sig = {'conv3_mfma_fwd_kernel<8, 1, false, false, true, false, false, true, 0, true, false, 2>@131072': 'da_conv3d_k3_fwd[32, 16, 2, 160, 192, 160, 16, 1]',
               'conv3_mfma_fwd_kernel<8, 1, false, true, true, false, false, true, 0, true, false, 2>': 'da_conv3d_k3_fwd_bnstats[32, 16, 2, 160, 192, 160, 16, 1]',
               'conv3_mfma_fwd_kernel<8, 1, false, false, true, false, false, true, 0, true, false, 2>@129024': 'da_conv3d_k3_dgrad[32, 16, 2, 160, 192, 160, 16, 1]',
               'conv3_split_wgrad_kernel<false, 2, false>': 'da_conv3d_k3_wgrad[32, 16, 2, 160, 192, 160, 16, 1]',
               'conv3_split_wgrad16_kernel<false>': 'da_conv3d_k3_wgrad[32, 16, 2, 160, 192, 160, 16, 1]'}      # (16-channel chunks: whole 64-B sectors per voxel)
    else:
        sig = {'conv3_mfma_fwd_kernel<16, 1, false, false, false, false, false, false, 0, false, false, 2>': 'da_conv3d_k3_fwd[32, 16, 2, 160, 192, 160, 16, 1]',
               'conv3_mfma_fwd_kernel<16, 1, false, true, false, false, false, false, 0, false, false, 2>': 'da_conv3d_k3_fwd_bnstats[32, 16, 2, 160, 192, 160, 16, 1]',
               'conv3_mfma_fwd_kernel<16, 3, false, false, false, false, false, false, 0, false, false, 2>': 'da_conv3d_k3_dgrad[32, 16, 2, 160, 192, 160, 16, 1]',
               'conv3_mfma_wgrad_kernel<16, 1, false, false, false, false, false, false>': 'da_conv3d_k3_wgrad[32, 16, 2, 160, 192, 160, 16, 1]'}
    # data gradient: collected in its own process (tools/pmc_conv.sh); per call = the sum over its conv kernels (split mode: two launches)
    dg_calls = 6.0                                   # bench_conv.py --iters 3: 3 warm-up + 3 timed calls
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = os.path.join(src, '%s_dgrad_%s.csv' % (layer, c))
        if not os.path.isfile(f):
            continue
        shutil.copyfile(f, os.path.join(dst, '%s_%s_dgrad_pmc_%s.csv' % (tag, layer, c.lower())))
        tot, names = 0.0, set()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            if k.startswith('conv3_mfma_fwd_kernel'):
                tot += float(r['Counter_Value']) * 1024.0
                names.add(k)
        per.setdefault('__dgrad__', {})[c] = tot / dg_calls
        per['__dgrad__']['names'] = ' + '.join(sorted(names))
    vox = 2 * 160 * 192 * 160
    alg = {'fwd': vox * (48 + 16) * 4, 'dgrad': vox * (16 + 48) * 4, 'wgrad': vox * (48 + 16) * 4 + 27 * 48 * 16 * 4}
    res = {'unit': 'bytes per launch', 'counters': 'FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes, KiB -> bytes)',
           'layer': '3x3x3 conv 48(=32+16 concat) -> 16, batch 2, 160x192x160 fp32', 'calls': {},
           'matrix_precision': {0: 'fp32', 1: 'bf16', 2: 'fp32_split'}.get(mode, 'fp32')}
    commit = '?'
    try:
        commit = open(os.path.join(src, 'commit.txt')).read().strip() or '?'
    except OSError:
        pass
    if commit == '?':          # the GPU box gets a snapshot without .git: the summary is folded where the history is
        try:
            import subprocess
            commit = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], cwd=os.path.dirname(os.path.abspath(__file__)), text=True).strip() + ' (summary folded at this commit; the CSVs were collected on the working-tree snapshot gpurun pushed at or shortly before it)'
        except Exception:
            pass
    res['commit'] = commit
    # calibration (tools/ubench/fetch_calib.hip): requested bytes / FETCH_SIZE for the two request shapes of the halo staging
    calib = {}
    fc = os.path.join(src, 'fetch_calib_FETCH_SIZE.csv')
    if os.path.isfile(fc):
        shutil.copyfile(fc, os.path.join(dst, '%s_fetch_calib_pmc_fetch_size.csv' % tag))
        # bytes of the 64-B sectors each kernel touches (= what has to cross the L2 - fabric interface): the 32-B-run shapes touch twice /
        # as many sector bytes as they request
        want = {'stream_b128_contig': float(1 << 30), 'stream_b128_half': float(1 << 29), 'stream_b128_run32<64>': float(1 << 30), 'stream_b128_run32<128>': float(1 << 29)}
        got = collections.defaultdict(list)
        for r in csv.DictReader(open(fc)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if k in want:
                got[k].append(float(r['Counter_Value']) * 1024.0)
        for k, v in got.items():
            v = v[1:] if len(v) > 1 else v
            m = sum(v) / len(v)
            calib[k] = {'sector_bytes_touched': want[k], 'fetch_size_bytes': m, 'requested_over_fetch_size': want[k] / m if m else None}
    res['fetch_size_calibration'] = calib
    if '__dgrad__' in per:
        sig = {k: v for k, v in sig.items() if 'dgrad' not in v}
        sig['__dgrad__'] = 'da_conv3d_k3_dgrad[32, 16, 2, 160, 192, 160, 16, 1]'
    for k, name in sig.items():
        if k not in per:
            continue
        f, w = per[k].get('FETCH_SIZE', 0.0), per[k].get('WRITE_SIZE', 0.0)
        a = alg[name.split('_')[3].split('[')[0]]            # 'fwd' also for da_conv3d_k3_fwd_bnstats
        # corrected fetch: the forward / weight-gradient kernels stage in1 (32-channel tensor: 64-B runs at a 128-B stride, "half" shape)
        # and in2 (16-channel tensor: contiguous, "contig" shape); the data gradient stages dy (16 channels: contiguous)
        fc_contig = (calib.get('stream_b128_contig') or {}).get('requested_over_fetch_size') or 1.0
        fc_half = (calib.get('stream_b128_half') or {}).get('requested_over_fetch_size') or 1.0
        # bytes staged per shape (in units of one 16-channel tensor): forward: in1 = 2 half-shape, in2 = 1 contiguous; weight gradient: the
        # same + dy = 1 contiguous; data gradient: dy = 1 contiguous.  raw = sum_s actual_s / factor_s with equal over-fetch ratios, so
        # actual = raw * sum_s B_s / sum_s (B_s / factor_s)
        b_half, b_contig = (0.0, 1.0) if 'dgrad' in name else ((2.0, 2.0) if 'wgrad' in name else (2.0, 1.0))
        if mode == 2 and 'wgrad16' not in k:      # (the 16-channel weight gradient stages the fp32 matrix mode's shapes: 64-B runs at a 128-B stride, contiguous rows)
            # split mode stages 8-channel chunks: in1 (32 channels) as 32-B runs at a 128-B stride, in2 / dy-as-input (16 channels) as 32-B runs at a
            # 64-B stride; the weight gradient's dY tile is staged in whole 64-B voxel rows (contiguous).  "half" below = the 128-B-stride
            # shape, "contig" = the 64-B-stride shape (+ the contiguous dY of the weight gradient, whose factor is folded in by weight)
            fcc = fc_contig
            fc_half = (calib.get('stream_b128_run32<128>') or {}).get('requested_over_fetch_size') or 1.0
            f64 = (calib.get('stream_b128_run32<64>') or {}).get('requested_over_fetch_size') or 2.0
            fc_contig = f64 if 'wgrad' not in name else 2.0 / (1.0 / f64 + 1.0 / fcc)
        share_half = b_half / (b_half + b_contig)
        fcorr = f * (b_half + b_contig) / (b_half / fc_half + b_contig / fc_contig)
        res['calls'][name] = {'kernel': per[k].get('names', k), 'fetch_bytes_raw': f, 'fetch_bytes': fcorr, 'write_bytes': w, 'traffic_bytes': fcorr + w,
                              'algorithmic_bytes': a, 'traffic_over_algorithmic': (fcorr + w) / a,
                              'fetch_correction': 'raw FETCH_SIZE x %.4f: %.0f %% of the staged bytes come from the 32-channel tensor (one 64-B sector per request, calibration '
                                                  'factor %.3f) and %.0f %% from 16-channel tensors (adjacent sectors merge into 128-B requests tallied as 64 B, factor %.3f); '
                                                  'raw = sum of actual / factor.  FETCH_SIZE counts L2 misses, Infinity-Cache hits included'
                                                  % (fcorr / f if f else 0.0, 100 * share_half, fc_half, 100 * (1 - share_half), fc_contig)}
    # third pass (SQ block): matrix-pipe occupancy.  SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles x (wave-level v_mfma_f32_16x16x4_f32 count),
    # summed over the 1024 SIMDs; divided by SIMDs and kernel duration it is the rate at which a SIMD's matrix pipe is busy, to be
    # read against the shader clock (2.4 GHz peak; ~1.95-2.0 GHz sustained under this load, DA_CLK probe in DESIGN.md 4.1).
    fsq = os.path.join(src, '%s_SQ.csv' % layer)
    if os.path.isfile(fsq):
        shutil.copyfile(fsq, os.path.join(dst, '%s_%s_pmc_sq_counters.csv' % (tag, layer)))
        sq = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fsq)):
            k0 = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            for k in (k0, k0 + '@' + r['Grid_Size']):
                sq[k][r['Counter_Name']].append(float(r['Counter_Value']))
                if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
                    sq[k]['duration_ns'].append(float(int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
        for k, name in sig.items():
            v = sq.get(k)
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
