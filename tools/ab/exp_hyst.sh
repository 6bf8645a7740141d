#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for ab in 0 1; do
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160 64,32,32,2,80,96,80; do
  echo "hysteresis off=$ab"
  DA_WG_ABLATE=$((ab*32)) DA_ABLATE=$((ab*8)) DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L 2>&1 | grep -v amdgpu.ids
done
done
done
