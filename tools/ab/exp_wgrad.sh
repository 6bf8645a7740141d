#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160 8,0,16,2,160,192,160 64,32,32,2,80,96,80 32,0,32,2,80,96,80 64,64,64,2,40,48,40 64,0,64,2,20,24,20; do
  DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what wgrad,wgradpro 2>&1 | grep -v amdgpu.ids
done
