#!/bin/bash
# Runs on the GPU box (through gpurun): bench lines + rocprofv3 kernel summary of the bf16 matrix mode -> gpurun_out/refresh_bf16/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh_bf16; rm -rf $O; mkdir -p $O
for w in seg reg joint; do
  timeout 600 python bench.py --workload $w --precision bf16 --no-cpu-baseline 2>&1 < /dev/null | grep '"metric"' | tail -1 > $O/bench_${w}_bf16.json
done
timeout 600 python bench.py --workload joint --precision bf16 --shape 192 224 192 --no-cpu-baseline 2>&1 < /dev/null | grep '"metric"' | tail -1 > $O/bench_joint_bf16_192x224x192.json
timeout 600 python bench.py --workload joint --shape 192 224 192 --no-cpu-baseline 2>&1 < /dev/null | grep '"metric"' | tail -1 > $O/bench_joint_fp32_192x224x192.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1 < /dev/null
f=$(ls $O/prof/*/*.db 2>/dev/null | head -1)
if [ -n "$f" ]; then python tools/rocpd_summary.py "$f" > $O/seg_bf16_kernel_stats.txt 2>&1 < /dev/null; python tools/rocpd_timeline.py "$f" --top 16 > $O/seg_bf16_timeline.txt 2>&1 < /dev/null; fi
rm -rf $O/prof
for l in 32,16,16,2,160,192,160 16,0,16,2,160,192,160 64,32,32,2,80,96,80 8,0,16,2,160,192,160; do
  DA_MATRIX_BF16=1 python tools/bench_conv.py --layer $l --what fwd,dgrad,wgrad --iters 10 2>&1 | grep -v amdgpu.ids >> $O/conv3d_layers_bf16.txt
done
cat $O/bench_*.json | cut -c1-260
