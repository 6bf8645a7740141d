#!/bin/bash
# GPU box: full gpu suite + rocprof kernel summaries of the reg and joint steps (scratch, for deciding what to work on).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/regprof; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/tests.txt
for w in reg joint; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$w -- python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/prof_$w.log 2>&1 < /dev/null
  f=$(ls $O/prof_$w/*/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/rocpd_summary.py "$f" --top 45 > $O/${w}_kernel_stats.txt 2>&1 < /dev/null
  rm -rf $O/prof_$w
done
