#!/usr/bin/env python
"""Fold the rocprofv3 --pmc passes of tools/pmc_conv.sh into profiles/rNN_pmc_traffic.json: HBM-side bytes per CALL of the dominant layer's
fwd + statistics / data gradient / weight gradient, keyed by the C-ABI call signature bench.py reports as `roofline.kernel`.

How a call's rows are found: tools/pmc_conv.sh runs ONE op per process, so every kernel row that is not a torch / runtime kernel belongs to that op
(conv kernels, their weight packs, tile tables, partial reductions), whatever the kernels are called at HEAD; per call = sum / calls of the process.
An op without rows is an ERROR (exit 2) -- round 4 lost the dominant kernel's record to a renamed template instantiation.

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE on gfx950 tallies every L2 -> fabric read
request at 64 B although 128-B requests exist (a wide coalesced stream reads exactly half), other widths "uncalibrated: calibrate on a known byte
count in your own access pattern".  Done here in two steps:
  1. tools/ubench/fetch_calib.hip sweeps a 1 GiB buffer once in four request shapes under the same passes.  On that KNOWN byte count (the 128-byte
     lines touched: the L2 fetches whole lines, see `want` below) the script checks which of
         raw FETCH_SIZE  |  request-size mix 32 n32 + 64 n64 + 128 n128  |  size-weighted 32 (DRAM_32B + GMI_32B + IO_32B)
     reproduces it (method accepted when every shape is within 3 %).  Round 5: both request-aware counters reproduce 1 GiB to four digits on all
     four shapes; raw FETCH_SIZE reads exactly half on all four (every request is 128 bytes, tallied as 64);
  2. the accepted method is read on the conv kernels themselves: fetch_bytes = that counter, fetch_correction = fetch_bytes / raw FETCH_SIZE --
     i.e. FETCH_SIZE corrected with THIS kernel's own request mix, not with an assumed mix of shapes.
If no method passes step 1 the script falls back to raw FETCH_SIZE x the calibration factor of the contiguous shape and says so.
WRITE_SIZE of the forward equals the algorithmic output exactly (checked below), so writes are not corrected.
Usage: python tools/pmc_summary.py gpurun_out/pmc profiles r05
       python tools/pmc_summary.py --averages counter_collection.csv KERNEL_SUBSTR [label]      (per-launch averages, for tools/ab/ab.sh pmc)"""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

CALLS_PER_PROCESS = 6.0          # tools/bench_conv.py --iters 3: max(3, iters) warm-up + iters timed calls
LAYER = 'conv3d_48to16'
OPS = {'fwdstats': 'da_conv3d_k3_fwd_bnstats[32, 16, 2, 160, 192, 160, 16, 1]',
       'dgrad': 'da_conv3d_k3_dgrad[32, 16, 2, 160, 192, 160, 16, 1]',
       'wgrad': 'da_conv3d_k3_wgrad[32, 16, 2, 160, 192, 160, 16, 1]'}
SKIP = ('at::', 'void at::', '__amd_rocclr', 'hipcub', 'rocprim', 'void rocprim', 'void hipcub')


def kname(r):
    return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def is_ours(name):
    return not any(name.startswith(s.replace('void ', '')) for s in SKIP)


def per_call(path):
    """{counter: bytes-or-count per call}, {kernel: launches per call} over our kernels of one one-op process."""
    tot, launches = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = kname(r)
        if not is_ours(k):
            continue
        tot[r['Counter_Name']] += float(r['Counter_Value'])
        launches[k].add(r['Dispatch_Id'])
    return {c: v / CALLS_PER_PROCESS for c, v in tot.items()}, {k: len(v) / CALLS_PER_PROCESS for k, v in launches.items()}


def calib_rows(path):
    """{kernel: {counter: mean per launch}} of the calibration microbenchmark (first launch of each kernel dropped: cold)."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[kname(r)][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for c, v in d.items()} for k, d in acc.items()}


def fetch_methods(cnt):
    """bytes by each method from a {counter: value} dict (missing counters -> method absent)."""
    m = {}
    if 'FETCH_SIZE' in cnt:
        m['fetch_size_raw'] = cnt['FETCH_SIZE'] * 1024.0
    if all(k in cnt for k in ('TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum', 'TCC_EA0_RDREQ_128B_sum')):
        n, n32, n128 = cnt['TCC_EA0_RDREQ_sum'], cnt['TCC_EA0_RDREQ_32B_sum'], cnt['TCC_EA0_RDREQ_128B_sum']
        m['request_mix'] = 32.0 * n32 + 128.0 * n128 + 64.0 * (n - n32 - n128)
    if 'TCC_EA0_RDREQ_DRAM_32B_sum' in cnt:
        m['size_weighted'] = 32.0 * (cnt['TCC_EA0_RDREQ_DRAM_32B_sum'] + cnt.get('TCC_EA0_RDREQ_GMI_32B_sum', 0.0) + cnt.get('TCC_EA0_RDREQ_IO_32B_sum', 0.0))
    return m


def averages(path, substr, label=''):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if substr in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(label, {k: '%.4g' % (sum(v) / len(v)) for k, v in acc.items()}, 'launches', max((len(v) for v in acc.values()), default=0))


def main():
    if sys.argv[1] == '--averages':
        return averages(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else '')
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    passes = ('FETCH_SIZE', 'WRITE_SIZE', 'RDMIX', 'RDW32', 'L1L2', 'SQ')
    mode = 0
    try:
        mode = int(open(os.path.join(src, 'matrix_mode.txt')).read().strip())
    except (OSError, ValueError):
        pass
    commit = '?'
    try:
        commit = open(os.path.join(src, 'commit.txt')).read().strip() or '?'
    except OSError:
        pass
    if commit == '?':          # the GPU box gets a snapshot without .git: the summary is folded where the history is
        try:
            commit = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], cwd=os.path.dirname(os.path.abspath(__file__)), text=True).strip() \
                + ' (summary folded at this commit; the CSVs were collected on the working-tree snapshot gpurun pushed at or shortly before it)'
        except Exception:
            pass
    # ---- step 1: which counter reproduces known byte counts
    # Known byte count of each calibration stream = the 128-BYTE LINES it touches: on gfx950 every L2 -> fabric read is a 128-byte request (RDMIX pass:
    # RDREQ_128B == RDREQ, no 32- / 64-byte requests at all), so a stream that uses only 32 or 64 bytes of every line (the `half` / `run32<128>` shapes:
    # one chunk of a 32-channel voxel) still moves the whole line.  All four streams sweep every line of the 1 GiB buffer exactly once.
    want = {'stream_b128_contig': float(1 << 30), 'stream_b128_half': float(1 << 30), 'stream_b128_run32<64>': float(1 << 30), 'stream_b128_run32<128>': float(1 << 30)}
    cal_cnt = collections.defaultdict(dict)
    for p in ('FETCH_SIZE', 'RDMIX', 'RDW32'):
        f = os.path.join(src, 'fetch_calib_%s.csv' % p)
        if os.path.isfile(f):
            shutil.copyfile(f, os.path.join(dst, '%s_fetch_calib_pmc_%s.csv' % (tag, p.lower())))
            for k, d in calib_rows(f).items():
                if k in want:
                    cal_cnt[k].update(d)
    calib, ok = {}, collections.defaultdict(list)
    for k, cnt in cal_cnt.items():
        ms = fetch_methods(cnt)
        calib[k] = {'sector_bytes_touched': want[k], 'by_method': {m: {'bytes': v, 'known_over_measured': want[k] / v if v else None} for m, v in ms.items()}}
        for m, v in ms.items():
            ok[m].append(bool(v) and abs(want[k] / v - 1.0) <= 0.03)
    accepted = [m for m in ('size_weighted', 'request_mix', 'fetch_size_raw') if ok.get(m) and len(ok[m]) == len(want) and all(ok[m])]
    method = accepted[0] if accepted else None
    contig = (calib.get('stream_b128_contig', {}).get('by_method', {}).get('fetch_size_raw') or {}).get('known_over_measured')
    vox = 2 * 160 * 192 * 160
    alg = {'fwdstats': vox * (48 + 16) * 4, 'dgrad': vox * (16 + 48) * 4, 'wgrad': vox * (48 + 16) * 4 + 27 * 48 * 16 * 4}
    res = {'unit': 'bytes per call', 'layer': '3x3x3 conv 48(=32+16 concat) -> 16, batch 2, 160x192x160 fp32', 'commit': commit,
           'matrix_precision': {0: 'fp32', 1: 'bf16', 2: 'fp32_split'}.get(mode, 'fp32'),
           'counters': 'rocprofv3 --pmc, one counter block per pass, one op per process (tools/pmc_conv.sh); KiB -> bytes',
           'fetch_method': method or 'fetch_size_raw x contiguous-stream factor (no counter reproduced the known byte counts within 3 %)',
           'fetch_size_calibration': calib, 'calls': {}}
    missing = []
    for op, name in OPS.items():
        cnt, launches = {}, {}
        for p in passes:
            f = os.path.join(src, '%s_%s_%s.csv' % (LAYER, op, p))
            if not os.path.isfile(f):
                continue
            shutil.copyfile(f, os.path.join(dst, '%s_%s_%s_pmc_%s.csv' % (tag, LAYER, op, p.lower())))
            c, l = per_call(f)
            cnt.update(c)
            launches = launches or l
            if p == 'SQ':          # kernel duration under the SQ pass, per call, conv kernels only
                dur = 0.0
                seen = set()
                for r in csv.DictReader(open(f)):
                    if is_ours(kname(r)) and r['Dispatch_Id'] not in seen:
                        seen.add(r['Dispatch_Id'])
                        dur += float(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
                cnt['duration_ns'] = dur / CALLS_PER_PROCESS
        if 'FETCH_SIZE' not in cnt or 'WRITE_SIZE' not in cnt:
            missing.append(op)
            continue
        ms = fetch_methods(cnt)
        raw = ms['fetch_size_raw']
        if method and method in ms:
            fetch, how = ms[method], "%s counter read on this call's own kernels (accepted on the calibration shapes)" % method
        else:
            fetch, how = raw * (contig or 2.0), 'raw FETCH_SIZE x %.3f (contiguous-stream calibration; the call mixes request sizes, so this is an upper estimate)' % (contig or 2.0)
        w = cnt['WRITE_SIZE'] * 1024.0
        rec = {'kernels_per_call': {k: round(v, 2) for k, v in sorted(launches.items())}, 'fetch_bytes_raw': raw, 'fetch_bytes': fetch, 'write_bytes': w,
               'traffic_bytes': fetch + w, 'algorithmic_bytes': alg[op], 'traffic_over_algorithmic': (fetch + w) / alg[op],
               'fetch_correction': fetch / raw if raw else None, 'fetch_correction_how': how,
               'fetch_bytes_by_method': ms,
               'note': 'FETCH_SIZE-class counters count L2 misses at the L2 - fabric interface, Infinity-Cache hits included'}
        if 'TCC_EA0_RDREQ_sum' in cnt:
            rec['read_requests'] = {k: cnt.get(k) for k in ('TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum', 'TCC_EA0_RDREQ_64B_sum', 'TCC_EA0_RDREQ_128B_sum')}
        if 'TCC_EA0_WRREQ_WRITE_DRAM_32B_sum' in cnt:
            rec['write_bytes_size_weighted'] = 32.0 * cnt['TCC_EA0_WRREQ_WRITE_DRAM_32B_sum']
        if 'TCP_TCC_READ_REQ_sum' in cnt:
            rq, acc_ = cnt['TCP_TCC_READ_REQ_sum'], cnt.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0.0)
            hit, miss = cnt.get('TCC_HIT_sum', 0.0), cnt.get('TCC_MISS_sum', 0.0)
            rec['memory_path'] = {'l1_to_l2_read_requests': rq, 'l1_to_l2_request_bytes_at_64B': 64.0 * rq,
                                  'l1_to_l2_mean_latency_cycles': cnt.get('TCP_TCC_READ_REQ_LATENCY_sum', 0.0) / rq if rq else None,
                                  'l1_hit_rate': 1.0 - rq / acc_ if acc_ else None, 'l2_hit_rate': hit / (hit + miss) if hit + miss else None,
                                  'l1_pending_stall_cycles_per_cu': cnt.get('TCP_PENDING_STALL_CYCLES_sum', 0.0) / 256.0}
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in cnt and cnt.get('duration_ns'):
            ghz = cnt['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cnt['duration_ns']
            rec['sq'] = {'mfma_busy_cycles': cnt['SQ_VALU_MFMA_BUSY_CYCLES'], 'duration_ms_under_pmc': cnt['duration_ns'] / 1e6, 'mfma_busy_ghz_per_simd': ghz,
                         'mfma_util_vs_2p4ghz_peak': ghz / 2.4, 'lds_bank_conflict_cycles': cnt.get('SQ_LDS_BANK_CONFLICT'), 'lds_active_cycles': cnt.get('SQ_LDS_IDX_ACTIVE'),
                         'wave_cycles': cnt.get('SQ_WAVE_CYCLES'), 'active_inst_any': cnt.get('SQ_ACTIVE_INST_ANY'), 'wait_inst_any': cnt.get('SQ_WAIT_INST_ANY'), 'wait_any': cnt.get('SQ_WAIT_ANY')}
        res['calls'][name] = rec
    with open(os.path.join(dst, '%s_pmc_traffic.json' % tag), 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))
    if missing:
        print('ERROR: no FETCH_SIZE / WRITE_SIZE rows for: %s' % ', '.join(missing), file=sys.stderr)
        sys.exit(2)


if __name__ == '__main__':
    main()
