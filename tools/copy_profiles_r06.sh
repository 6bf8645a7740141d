#!/bin/bash
# Copy the summaries tools/refresh_profiles_r06.sh wrote under gpurun_out/r06p/ (scratch) to their tracked names under profiles/.
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r06p; P=profiles
for w in seg reg joint joint_smooth; do
  for k in calls kernel_stats timeline; do
    [ -f $S/${w}_$k.txt ] && cp $S/${w}_$k.txt $P/r06_${w}_160x192x160_$k.txt
  done
  for e in txt json; do [ -f $S/step_traffic_$w.$e ] && cp $S/step_traffic_$w.$e $P/r06_step_traffic_$w.$e; done
done
[ -f $S/hbm_bound_calls.txt ] && cp $S/hbm_bound_calls.txt $P/r06_hbm_bound_calls_160x192x160.txt
for f in conv3d_48to16_kernel_stats conv_layers_isolated conv_zero_operands gather_kernels ablate_step_seg pointwise_calls concurrency_probe deconv_bwd_ablate; do [ -f $S/$f.txt ] && cp $S/$f.txt $P/r06_$f.txt; done
for f in $S/bench_*.json; do [ -f $f ] && cp $f $P/r06_$(basename $f); done
ls $P | grep -c r06_
