#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_split.py -q -x 2>&1 | tail -2
for rep in 1 2; do
for v in 0 1; do
  echo "DA_WG16=$v"
  DA_WG16=$v python bench.py --workload seg --no-cpu-baseline --no-extra 2>&1 | grep '"metric"' > /tmp/b.json; python tools/bench_brief.py /tmp/b.json | head -1
done
done
for v in 0 1; do
  echo "joint DA_WG16=$v"
  DA_WG16=$v python bench.py --workload joint --no-cpu-baseline --no-extra 2>&1 | grep '"metric"' > /tmp/b.json; python tools/bench_brief.py /tmp/b.json | head -1
done
