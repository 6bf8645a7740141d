#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py tests/test_gpu_random_shapes.py -q -x 2>&1 | tail -3
for v in 1 0 1 0; do
  echo "DA_NO_FEWCIN_VALU=$v"
  DA_NO_FEWCIN_VALU=$v python bench.py --workload seg --no-cpu-baseline --no-extra 2>&1 | grep '"metric"' > /tmp/b.json; python tools/bench_brief.py /tmp/b.json | head -1
  DA_NO_FEWCIN_VALU=$v python bench.py --workload reg --no-cpu-baseline --no-extra 2>&1 | grep '"metric"' > /tmp/b.json; python tools/bench_brief.py /tmp/b.json | head -1
done
DA_NO_FEWCIN_VALU=0 python tools/step_calls.py seg 2>&1 | grep "1, 0, 2, 160, 192, 160, 8" 
DA_NO_FEWCIN_VALU=1 python tools/step_calls.py seg 2>&1 | grep "1, 0, 2, 160, 192, 160, 8" 
