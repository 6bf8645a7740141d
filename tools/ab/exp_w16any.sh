#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for any in 0 1; do
for L in 64,32,32,2,80,96,80 128,64,64,2,40,48,40 64,0,64,2,20,24,20; do
  echo "DA_WG16_ANY=$any"
  DA_WG16_ANY=$any DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what wgrad,wgradpro 2>&1 | grep -v amdgpu.ids
done
done
done
