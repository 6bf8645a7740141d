#!/bin/bash
# Copy the summaries tools/refresh_profiles_r05.sh wrote under gpurun_out/r05p/ (scratch) to their tracked names under profiles/.
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r05p; P=profiles
for w in seg reg joint; do
  for k in calls kernel_stats timeline; do
    [ -f $S/${w}_$k.txt ] && cp $S/${w}_$k.txt $P/r05_${w}_160x192x160_$k.txt
  done
done
cp $S/hbm_bound_calls.txt $P/r05_hbm_bound_calls_160x192x160.txt
for f in conv3d_48to16_kernel_stats conv_layers_isolated gather_kernels; do cp $S/$f.txt $P/r05_$f.txt; done
for f in $S/bench_*.json; do cp $f $P/r05_$(basename $f); done
ls $P | grep -c r05_
