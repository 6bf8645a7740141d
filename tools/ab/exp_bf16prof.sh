#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bf16prof; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --workload seg --precision bf16_storage --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/prof.log 2>&1 < /dev/null
f=$(ls $O/prof/*/*.db | head -1)
python tools/rocpd_summary.py "$f" --top 30 > $O/seg_kernel_stats.txt 2>&1
python tools/rocpd_timeline.py "$f" > $O/seg_timeline.txt 2>&1
rm -rf $O/prof
PRECISION=bf16_storage python tools/step_calls.py seg 2>&1 | grep -v amdgpu.ids | head -40 > $O/seg_calls.txt
