#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel summary + two-stream timeline + per-C-ABI-call table of one workload.
#   tools/prof_step.sh <seg|reg|joint> <tag>   ->  gpurun_out/r04/<tag>_{kernel_stats,timeline,calls}.txt
w=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${ROUND:-r04}; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -- python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/prof_$tag.log 2>&1 < /dev/null
f=$(ls $O/prof_$tag/*/*.db 2>/dev/null | head -1)
if [ -n "$f" ]; then
  python tools/rocpd_summary.py "$f" --top 60 > $O/${tag}_kernel_stats.txt 2>&1 < /dev/null
  python tools/rocpd_timeline.py "$f" > $O/${tag}_timeline.txt 2>&1 < /dev/null
  python tools/rocpd_timeline.py "$f" --dump --dump-all > $O/${tag}_step_dump.txt 2>&1 < /dev/null
fi
rm -rf $O/prof_$tag
timeout 600 python tools/step_calls.py $w 2>&1 | grep -v amdgpu.ids > $O/${tag}_calls.txt
