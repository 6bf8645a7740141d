#!/bin/bash
# GPU box (via gpurun): fabric-side bytes of a WHOLE training step, per kernel (VERDICT r5 item 2 / "missing" 9).
# Two rocprofv3 --pmc passes over bench.py's workload (one counter block per pass, MI355X_MICROARCH.md): reads from the size-weighted request counters
# (32-byte units to DRAM / GMI / IO: the method profiles/r05_pmc_traffic.json found to reproduce known byte counts to four digits; raw FETCH_SIZE reads
# half), writes from WRITE_SIZE (KiB; equals the algorithmic output of the convolutions exactly).  tools/pmc_step_summary.py folds them per step.
#   tools/pmc_step.sh <seg|reg|joint> [steps]     ->  gpurun_out/pmc_step/<workload>_{rd,wr}.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
W=${1:-seg}; K=${2:-3}
O=gpurun_out/pmc_step; mkdir -p $O
pass() {   # pass NAME "COUNTERS"
  rm -rf $O/tmp
  timeout 900 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/tmp -- python bench.py --workload $W --steps $K --warmup 2 --no-cpu-baseline --no-extra --no-profile > $O/${W}_$1.log 2>&1 < /dev/null
  f=$(ls $O/tmp/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" $O/${W}_$1.csv; else echo "pass $1: no counter csv"; tail -5 $O/${W}_$1.log; fi
  rm -rf $O/tmp
}
pass rd "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum TCC_EA0_RDREQ_sum"
pass wr "WRITE_SIZE"
python tools/pmc_step_summary.py $O $W
