#!/bin/bash
# GPU box (via gpurun): SQ counters of one conv call (two passes of <= 8 SQ counters).  usage: tools/pmc_sq.sh <mode> <layer> <what>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
M=${1:-2}; L=${2:-32,16,16,2,160,192,160}; W=${3:-fwd}
O=gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  DA_MATRIX_MODE=$M timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -- python tools/bench_conv.py --layer $L --what $W --iters 3 > $O/p$i.log 2>&1 < /dev/null
  f=$(ls $O/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" $O/pass$i.csv; fi
  rm -rf $O/p$i
done
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc_sq/pass*.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if 'conv3' not in k: continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        print(k)
        for c, v in d.items():
            v = v[1:] if len(v) > 1 else v
            print('   %-28s %16.0f' % (c, sum(v) / len(v)))
PY
