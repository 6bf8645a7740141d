#!/bin/bash
# Runs on the GPU box (through gpurun): round-3 evidence -> gpurun_out/r03p/ (copy what is to be judged into profiles/).
#   bench lines (default run = seg headline + reg + joint + configs[4]-shape legs + fp32 A/B + CPU oracle + full-size parity; fp32 MFMA; bf16;
#   graph mode; host floor), rocprofv3 kernel summaries + two-stream timelines + per-C-ABI-call tables of the three workloads, the
#   HBM-bound call table, isolated conv layers incl. the native stride-2 and folded up-sampling layers.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03p; rm -rf $O; mkdir -p $O
git rev-parse --short HEAD > $O/commit.txt 2>/dev/null || true
timeout 1200 python bench.py > $O/bench_default.log 2>&1 < /dev/null
grep '"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.log 2>&1 < /dev/null
grep '"metric"' $O/bench_driver_style.log | tail -1 > $O/bench_driver_style_steps20.json
timeout 600 python bench.py --no-cpu-baseline --precision fp32 > $O/bench_fp32_mfma.log 2>&1 < /dev/null
grep '"metric"' $O/bench_fp32_mfma.log | tail -1 > $O/bench_fp32_mfma.json
timeout 600 python bench.py --no-cpu-baseline --precision bf16 > $O/bench_bf16.log 2>&1 < /dev/null
grep '"metric"' $O/bench_bf16.log | tail -1 > $O/bench_bf16.json
timeout 600 python bench.py --no-cpu-baseline --precision bf16_storage > $O/bench_bf16_storage.log 2>&1 < /dev/null
grep '"metric"' $O/bench_bf16_storage.log | tail -1 > $O/bench_bf16_storage.json
timeout 600 python bench.py --no-cpu-baseline --no-extra --precision bf16_storage --shape 192 224 192 --steps 6 --warmup 2 > $O/bench_bf16_storage_192x224x192.log 2>&1 < /dev/null
grep '"metric"' $O/bench_bf16_storage_192x224x192.log | tail -1 > $O/bench_bf16_storage_192x224x192.json
for w in seg reg joint; do
  PRECISION=bf16_storage timeout 600 python tools/step_calls.py $w 2>&1 | grep -v amdgpu.ids > $O/${w}_bf16_storage_calls.txt
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_segh -- python bench.py --workload seg --precision bf16_storage --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/prof_segh.log 2>&1 < /dev/null
f=$(ls $O/prof_segh/*/*.db 2>/dev/null | head -1)
if [ -n "$f" ]; then
  python tools/rocpd_summary.py "$f" --top 40 > $O/seg_bf16_storage_kernel_stats.txt 2>&1 < /dev/null
  python tools/rocpd_timeline.py "$f" > $O/seg_bf16_storage_timeline.txt 2>&1 < /dev/null
fi
rm -rf $O/prof_segh
echo "# bf16 activation storage (DA_MATRIX_MODE=1, --bf16-storage) and the ablation of the 48 -> 16 forward / data gradient (DA_ABLATE: 1 no staging loads, 2 no epilogue, 4 no LDS writes + barriers)" > $O/conv_layers_bf16_storage.txt
bash tools/ab/exp2.sh >> $O/conv_layers_bf16_storage.txt 2>&1
echo "# split-arithmetic ablation: the shipped library vs a build whose da_split3 does no arithmetic (tools/ab/libda_fakesplit.so, -DDA_FAKE_SPLIT; wrong results, same data volume)" > $O/conv_layers_fake_split.txt
if [ -f tools/ab/libda_fakesplit.so ]; then bash tools/ab/exp1.sh >> $O/conv_layers_fake_split.txt 2>&1; fi
timeout 600 python bench.py --no-cpu-baseline --graph > $O/bench_graph.log 2>&1 < /dev/null
grep '"metric"' $O/bench_graph.log | tail -1 > $O/bench_graph.json
timeout 600 python bench.py --no-cpu-baseline --shape 32 32 32 --steps 20 --warmup 5 --no-profile > $O/bench_host_floor_32.log 2>&1 < /dev/null
grep '"metric"' $O/bench_host_floor_32.log | tail -1 > $O/bench_host_floor_32.json
for w in seg reg joint; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$w -- python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/prof_$w.log 2>&1 < /dev/null
  f=$(ls $O/prof_$w/*/*.db 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python tools/rocpd_summary.py "$f" --top 60 > $O/${w}_kernel_stats.txt 2>&1 < /dev/null
    python tools/rocpd_timeline.py "$f" > $O/${w}_timeline.txt 2>&1 < /dev/null
  fi
  rm -rf $O/prof_$w
  timeout 600 python tools/step_calls.py $w 2>&1 | grep -v amdgpu.ids > $O/${w}_calls.txt
done
timeout 600 python tools/bench_losses.py 2>&1 | grep -v amdgpu.ids | grep -v '^\[' > $O/hbm_bound_calls.txt
rm -f $O/conv_layers_isolated.txt
echo "# DA_MATRIX_MODE=2 (fp32_split), tools/bench_conv.py --layer C1,C2,Cout,N,D,H,W" >> $O/conv_layers_isolated.txt
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160 8,0,16,2,160,192,160 64,32,32,2,80,96,80 64,64,64,2,40,48,40; do
  DA_MATRIX_MODE=2 timeout 600 python tools/bench_conv.py --layer $L 2>&1 | grep -v amdgpu.ids >> $O/conv_layers_isolated.txt
done
DA_MATRIX_MODE=2 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_conv -- python tools/bench_conv.py --layer 32,16,16,2,160,192,160 --iters 5 > $O/prof_conv.log 2>&1 < /dev/null
f=$(ls $O/prof_conv/*/*.db 2>/dev/null | head -1)
if [ -n "$f" ]; then python tools/rocpd_summary.py "$f" > $O/conv3d_48to16_kernel_stats.txt 2>&1 < /dev/null; fi
rm -rf $O/prof_conv
timeout 300 python tools/bench_warp.py 2>&1 | grep -v amdgpu.ids > $O/gather_kernels_now.txt
rm -f $O/*.log
ls -la $O
