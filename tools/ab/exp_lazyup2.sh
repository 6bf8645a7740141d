#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  echo "DA_LAZY_BN_UPSAMPLER=$v"
  DA_LAZY_BN_UPSAMPLER=$v python bench.py --no-cpu-baseline 2>&1 | grep '"metric"' > /tmp/b.json; python tools/bench_brief.py /tmp/b.json | grep -v "bf16\|native\|    da_\|all conv"
done
