#!/bin/bash
# Memory-path counters of the split weight gradient (one block of counters per pass): where do its staging loads wait?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export DA_MATRIX_MODE=2
O=gpurun_out/wgpmc; rm -rf $O; mkdir -p $O
i=0
for set in "TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  for L in 32,16,16,2,160,192,160 8,0,16,2,160,192,160; do
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p -- python tools/bench_conv.py --layer $L --what wgrad --iters 2 > $O/p$i.log 2>&1 < /dev/null
    f=$(ls $O/p/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then python - "$f" "$L" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'split_wgrad' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: '%.4g' % (sum(v) / len(v)) for k, v in acc.items()}, 'launches', max((len(v) for v in acc.values()), default=0))
PY
    else echo "pass $i $L: no csv"; tail -3 $O/p$i.log; fi
    rm -rf $O/p
  done
done
