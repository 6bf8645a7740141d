#!/bin/bash
# Upper bound of a two-term split (2 planes, 3 products): tools/ab/libda_fakesplit2.so (-DDA_FAKE_SPLIT2; wrong results) vs the shipped library.
cd $GRAFT_REPO_ROOT
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160 64,32,32,2,80,96,80 64,64,64,2,40,48,40; do
 for lib in "" tools/ab/libda_fakesplit2.so; do
  echo "== layer $L lib=${lib:-shipped}"
  DA_LIB=$lib DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what fwdstats,dgrad,wgrad 2>&1 | grep -v amdgpu.ids
 done
done
for lib in "" tools/ab/libda_fakesplit2.so; do
  echo "== bench.py seg step lib=${lib:-shipped}"
  DA_LIB=$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-profile 2>&1 | grep -v amdgpu.ids | grep -o '"ms_per_step": [0-9.]*' | head -1
done
