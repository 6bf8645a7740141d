#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160; do
for v in 0 1; do
  echo "== layer $L DA_FWD_WG3=$v"
  DA_FWD_WG3=$v DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what fwd,fwdstats,dgrad 2>&1 | grep -v amdgpu.ids
done
echo "== layer $L DA_NO_PAIR=1"
DA_NO_PAIR=1 DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what fwd,fwdstats,dgrad 2>&1 | grep -v amdgpu.ids
done
