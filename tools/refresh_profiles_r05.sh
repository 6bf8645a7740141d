#!/bin/bash
# Runs on the GPU box (through gpurun): round-5 evidence -> gpurun_out/r05p/ (tools/copy_profiles_r04.sh copies what is to be judged into profiles/).
#   bench lines (default run; driver-style; fp32 MFMA; bf16 storage), rocprofv3 kernel summaries + two-stream timelines + per-C-ABI-call tables
#   of the three workloads, the HBM-bound call table, isolated conv layers, the 48 -> 16 layer's kernel stats.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05p; rm -rf $O; mkdir -p $O
timeout 1500 python bench.py > $O/bench_default.log 2>&1 < /dev/null
grep '"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.log 2>&1 < /dev/null
grep '"metric"' $O/bench_driver_style.log | tail -1 > $O/bench_driver_style_steps20.json
timeout 600 python bench.py --no-cpu-baseline --no-extra --precision fp32 > $O/bench_fp32_mfma.log 2>&1 < /dev/null
grep '"metric"' $O/bench_fp32_mfma.log | tail -1 > $O/bench_fp32_mfma.json
timeout 600 python bench.py --no-cpu-baseline --precision bf16_storage > $O/bench_bf16_storage.log 2>&1 < /dev/null
grep '"metric"' $O/bench_bf16_storage.log | tail -1 > $O/bench_bf16_storage.json
for w in seg reg joint; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$w -- python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/prof_$w.log 2>&1 < /dev/null
  f=$(ls $O/prof_$w/*/*.db 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python tools/rocpd_summary.py "$f" --top 60 > $O/${w}_kernel_stats.txt 2>&1 < /dev/null
    python tools/rocpd_timeline.py "$f" $([ $w = joint ] && echo --adam-per-step 2) > $O/${w}_timeline.txt 2>&1 < /dev/null
  fi
  rm -rf $O/prof_$w
  timeout 600 python tools/step_calls.py $w 2>&1 | grep -v amdgpu.ids > $O/${w}_calls.txt
done
timeout 600 python tools/bench_losses.py 2>&1 | grep -v amdgpu.ids | grep -v '^\[' > $O/hbm_bound_calls.txt
echo "# DA_MATRIX_MODE=2 (fp32_split: two-term fp16 split), tools/bench_conv.py --layer C1,C2,Cout,N,D,H,W" > $O/conv_layers_isolated.txt
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160 8,0,16,2,160,192,160 64,32,32,2,80,96,80 32,0,32,2,80,96,80 64,64,64,2,40,48,40; do
  DA_MATRIX_MODE=2 timeout 600 python tools/bench_conv.py --layer $L 2>&1 | grep -v amdgpu.ids >> $O/conv_layers_isolated.txt
done
DA_MATRIX_MODE=2 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_conv -- python tools/bench_conv.py --layer 32,16,16,2,160,192,160 --iters 5 > $O/prof_conv.log 2>&1 < /dev/null
f=$(ls $O/prof_conv/*/*.db 2>/dev/null | head -1)
if [ -n "$f" ]; then python tools/rocpd_summary.py "$f" > $O/conv3d_48to16_kernel_stats.txt 2>&1 < /dev/null; fi
rm -rf $O/prof_conv
timeout 300 python tools/bench_warp.py 2>&1 | grep -v amdgpu.ids > $O/gather_kernels.txt
bash tools/pmc_conv.sh 2 > $O/pmc.log 2>&1
rm -f $O/bench_*.log $O/prof_*.log
ls -la $O
