#!/bin/bash
# GPU box (via gpurun): SQ counters of the bf16-storage conv kernels and of the split weight gradient (three passes of <= 8 SQ counters each) -> gpurun_out/pmc_sq2/summary.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_sq2; rm -rf $O; mkdir -p $O
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVES"
run() {   # tag, matrix mode, extra bench_conv args
  local tag=$1 mode=$2; shift 2
  local i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    DA_MATRIX_MODE=$mode timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p -- python tools/bench_conv.py --layer 32,16,16,2,160,192,160 --iters 3 "$@" > $O/run.log 2>&1 < /dev/null
    f=$(ls $O/p/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then cp "$f" $O/${tag}_pass$i.csv; fi
    rm -rf $O/p
  done
}
run split 2 --what fwdstats,dgrad,wgrad
run bf16h 1 --what fwdstats,dgrad,wgrad --bf16-storage
python - <<'PY' > gpurun_out/pmc_sq2/summary.txt
import csv, collections, glob
for tag in ('split', 'bf16h'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob('gpurun_out/pmc_sq2/%s_pass*.csv' % tag)):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            if 'conv3' not in k: continue
            agg[k + ' grid ' + r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==== %s (48 -> 16, 2 x 160 x 192 x 160; mean per launch, first launch dropped)' % tag)
    for k, d in agg.items():
        print(k)
        m = {c: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for c, v in d.items()}
        for c in sorted(m):
            print('   %-28s %16.0f' % (c, m[c]))
        if m.get('SQ_BUSY_CYCLES') and m.get('SQ_WAVE_CYCLES'):
            wc = m['SQ_WAVE_CYCLES']
            print('   -> per wave-cycle: active %.3f  wait_inst %.3f  wait_any %.3f ; MFMA busy / (GUI_ACTIVE x 1024 SIMDs) %.3f ; LDS active / (GUI_ACTIVE x 256 CUs) %.3f ; bank-conflict share of LDS cycles %.3f'
                  % (m.get('SQ_ACTIVE_INST_ANY', 0) / wc, m.get('SQ_WAIT_INST_ANY', 0) / wc, m.get('SQ_WAIT_ANY', 0) / wc,
                     m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (m.get('GRBM_GUI_ACTIVE', 1) * 1024.0), m.get('SQ_LDS_IDX_ACTIVE', 0) / (m.get('GRBM_GUI_ACTIVE', 1) * 256.0),
                     m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
cat gpurun_out/pmc_sq2/summary.txt | head -150
