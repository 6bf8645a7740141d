#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for n in 1 2 3; do
  echo "DA_SIDE_STREAMS=$n"
  for w in seg reg joint; do
    DA_SIDE_STREAMS=$n python bench.py --workload $w --no-cpu-baseline --no-extra 2>&1 | grep '"metric"' > /tmp/b.json; python tools/bench_brief.py /tmp/b.json | head -1
  done
done
done
DA_SIDE_STREAMS=2 python -m pytest tests/test_gpu_nets.py tests/test_gpu_dp.py -q -x 2>&1 | tail -2
