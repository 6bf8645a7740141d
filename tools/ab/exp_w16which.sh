#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 1 0; do
DA_WG16=$v rocprofv3 --kernel-trace --stats -d /tmp/pw$v -- python tools/ab/w16_compare.py /tmp/w16_$v.npz > /tmp/pw$v.log 2>&1
f=$(ls /tmp/pw$v/*/*.db | head -1)
echo "DA_WG16=$v"; python tools/rocpd_summary.py "$f" --top 12 2>&1 | grep -i "wgrad" | cut -c1-120
done
