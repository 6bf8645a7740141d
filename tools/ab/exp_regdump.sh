#!/bin/bash
# GPU box: reg step bench line + every dispatch of the last reg step (start / end / queue) to see what overlaps what.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${WL:-reg}dump; rm -rf $O; mkdir -p $O

timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --workload ${WL:-reg} --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/prof.log 2>&1 < /dev/null
f=$(ls $O/prof/*/*.db | head -1)
python tools/rocpd_timeline.py "$f" --dump --dump-all > $O/dump.txt 2>&1
python tools/rocpd_timeline.py "$f" > $O/timeline.txt 2>&1
rm -rf $O/prof
