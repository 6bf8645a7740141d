#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for np in 0 1; do
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160; do
  echo "DA_NO_PAIR=$np"
  DA_NO_PAIR=$np DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what fwd,fwdstats,dgrad,fwdpro 2>&1 | grep -v amdgpu.ids
done
done
done
