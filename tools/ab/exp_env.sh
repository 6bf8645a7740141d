#!/bin/bash
# A/B of environment switches on the seg step: tools/ab/exp_env.sh "VAR=1" "VAR2=1" ...
cd $GRAFT_REPO_ROOT
for e in "" "$@"; do
  for rep in 1 2; do
  echo "== env: ${e:-default}"
  env $e timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-profile 2>&1 | grep -v amdgpu.ids | grep -o '"ms_per_step": [0-9.]*' | head -1
  done
done
