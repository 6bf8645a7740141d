#!/bin/bash
# DA_ABLATE on the split-mode forward / data gradient: 1 no staging loads, 2 no epilogue, 4 no LDS writes + barriers
cd $GRAFT_REPO_ROOT
for L in 32,16,16,2,160,192,160 64,32,32,2,80,96,80; do
for ab in 0 1 2 4 5 7; do
  echo "== layer $L DA_ABLATE=$ab"
  DA_ABLATE=$ab DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what fwd,fwdstats,dgrad 2>&1 | grep -v amdgpu.ids
done
done
