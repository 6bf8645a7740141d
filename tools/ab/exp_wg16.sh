#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_split.py -q -x 2>&1 | tail -8
for on in 0 1 0 1; do
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160 32,0,32,2,80,96,80 64,64,64,2,40,48,40; do
  echo "DA_WG16=$on"
  DA_WG16=$on DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what wgrad,wgradpro 2>&1 | grep -v amdgpu.ids
done
done
