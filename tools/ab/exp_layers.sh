#!/bin/bash
# per-layer timings of the shipped library (split mode) + a short seg step
cd $GRAFT_REPO_ROOT
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160 8,0,16,2,160,192,160 64,32,32,2,80,96,80 32,0,32,2,80,96,80 64,64,64,2,40,48,40 128,0,64,2,40,48,40; do
  DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what fwd,fwdstats,dgrad,wgrad 2>&1 | grep -v amdgpu.ids
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_quick.json
python tools/bench_brief.py gpurun_out/bench_quick.json
