#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1; do
  echo "DA_LAZY_BN_UPSAMPLER=$v"
  for w in seg joint; do
  DA_LAZY_BN_UPSAMPLER=$v python bench.py --workload $w --no-cpu-baseline --no-extra 2>&1 | grep '"metric"' > /tmp/b.json; python tools/bench_brief.py /tmp/b.json | head -1
  done
done
done
