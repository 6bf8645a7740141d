#!/bin/bash
# DA_WG_ABLATE on the split-mode weight gradient: 1 no staging loads, 2 no fragment reads + MFMAs, 4 no maxima / LDS writes / barriers, 16 no rescale
cd $GRAFT_REPO_ROOT
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160; do
for ab in 0 1 2 4 16 5 3 6 7; do
  echo "== layer $L DA_WG_ABLATE=$ab"
  DA_WG_ABLATE=$ab DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what wgrad 2>&1 | grep -v amdgpu.ids | grep -i wgrad
done
done
