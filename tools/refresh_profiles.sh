#!/bin/bash
# Runs on the GPU box (through gpurun): bench lines + rocprofv3 kernel summaries for the three workloads -> gpurun_out/refresh/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
for w in seg reg joint; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.log 2>&1 < /dev/null
  grep '"metric"' $O/bench_$w.log | tail -1 > $O/bench_$w.json
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$w -- python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_$w.log 2>&1 < /dev/null
  f=$(ls $O/prof_$w/*/*.db 2>/dev/null | head -1)
  if [ -n "$f" ]; then python tools/rocpd_summary.py "$f" > $O/${w}_kernel_stats.txt 2>&1 < /dev/null; fi
  rm -rf $O/prof_$w
done
cat $O/bench_*.json | cut -c1-400
