#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_split.py -q -x 2>&1 | tail -3
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160 64,32,32,2,80,96,80 32,0,32,2,80,96,80; do
  DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L 2>&1 | grep -v amdgpu.ids
done
