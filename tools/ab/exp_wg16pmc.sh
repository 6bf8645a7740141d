#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export DA_MATRIX_MODE=2
O=gpurun_out/wg16pmc; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_WAVE32_LDS SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for w16 in 1 0; do
    DA_WG16=$w16 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p -- python tools/bench_conv.py --layer 32,16,16,2,160,192,160 --what wgrad --iters 2 > $O/p$i.log 2>&1 < /dev/null
    f=$(ls $O/p/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then python - "$f" "w16=$w16" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'split_wgrad' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: '%.4g' % (sum(v) / len(v)) for k, v in acc.items()})
PY
    else echo "pass $i w16=$w16: no csv"; tail -2 $O/p$i.log; fi
    rm -rf $O/p
  done
done
