#!/bin/bash
# Copy the summaries tools/refresh_profiles_r03.sh wrote under gpurun_out/r03p/ (scratch) to their tracked names under profiles/.
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r03p; P=profiles
for w in seg reg joint; do
  for k in calls kernel_stats timeline bf16_storage_calls bf16_storage_kernel_stats bf16_storage_timeline; do
    [ -f $S/${w}_$k.txt ] && cp $S/${w}_$k.txt $P/r03_${w}_160x192x160_$k.txt
  done
done
cp $S/hbm_bound_calls.txt $P/r03_hbm_bound_calls_160x192x160.txt
for f in conv3d_48to16_kernel_stats conv_layers_bf16_storage conv_layers_fake_split conv_layers_isolated; do cp $S/$f.txt $P/r03_$f.txt; done
cp $S/gather_kernels_now.txt $P/r03_gather_kernels_final.txt
for f in $S/bench_*.json; do cp $f $P/r03_$(basename $f); done
ls $P | grep -c r03_
