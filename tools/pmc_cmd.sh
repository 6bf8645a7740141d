#!/bin/bash
# GPU box (via gpurun): SQ counters (three passes of <= 9 counters) of the kernels whose name contains <match>, for an arbitrary python command.
#   tools/pmc_cmd.sh <tag> <match> <python args...>  ->  gpurun_out/r03/<tag>_pmc_sq.txt
tag=$1; match=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O; W=$O/pmc_$tag; rm -rf $W; mkdir -p $W
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $W/p$i -- python "$@" > $W/p$i.log 2>&1 < /dev/null
  f=$(ls $W/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" $W/pass$i.csv; fi
  rm -rf $W/p$i
done
MATCH="$match" WDIR="$W" python - > $O/${tag}_pmc_sq.txt <<'PY'
import csv, collections, glob, os
m = os.environ['MATCH']
for f in sorted(glob.glob(os.environ['WDIR'] + '/pass*.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if m not in k: continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        print(k)
        for c, v in d.items():
            v = v[1:] if len(v) > 1 else v
            print('   %-28s %16.0f   (%d launches)' % (c, sum(v) / len(v), len(v)))
PY
rm -rf $W
