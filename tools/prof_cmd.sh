#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel summary of an arbitrary python command.
#   tools/prof_cmd.sh <tag> <python args...>   ->  gpurun_out/r03/<tag>_kernel_stats.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -- python "$@" > $O/prof_$tag.log 2>&1 < /dev/null
f=$(ls $O/prof_$tag/*/*.db 2>/dev/null | head -1)
if [ -n "$f" ]; then
  python tools/rocpd_summary.py "$f" --top 40 > $O/${tag}_kernel_stats.txt 2>&1 < /dev/null
else
  echo "no rocprofv3 database under $O/prof_$tag" > $O/${tag}_kernel_stats.txt
fi
rm -rf $O/prof_$tag
