#!/bin/bash
cd $GRAFT_REPO_ROOT
L=32,16,16,2,160,192,160
for lib in "" tools/ab/libda_lb2.so tools/ab/libda_lb3.so; do
for ab in 0 7; do
  echo "== lib=${lib:-shipped} DA_ABLATE=$ab"
  DA_LIB=$lib DA_ABLATE=$ab DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what fwd,fwdstats,dgrad 2>&1 | grep -v amdgpu.ids
done
done
