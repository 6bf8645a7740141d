#!/bin/bash
# Runs on the GPU box (through gpurun): round-6 evidence -> gpurun_out/r06p/ (tools/copy_profiles_r06.sh copies what is to be judged into profiles/).
#   bench lines (default run; driver-style; fp32 MFMA; bf16 storage), rocprofv3 kernel summaries + two-stream timelines + per-C-ABI-call tables of the
#   three workloads, WHOLE-STEP fabric traffic of the three workloads (tools/pmc_step.sh), the HBM-bound call table, gather kernels, isolated conv
#   layers (shipped kernels and the opt-in conv3d_fwdsp.hip forms), the in-step ablation of the forward / data-gradient kernel, the zero-operand
#   (power) runs, the 48 -> 16 layer's kernel stats and PMC traffic.
# Parts: tools/refresh_profiles_r06.sh [all | bench | prof | traffic | kernels | ablate | pmc]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
PART=${1:-all}
O=gpurun_out/r06p; mkdir -p $O
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
if want bench; then
  timeout 1500 python bench.py > $O/bench_default.log 2>&1 < /dev/null
  grep '"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
  timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.log 2>&1 < /dev/null
  grep '"metric"' $O/bench_driver_style.log | tail -1 > $O/bench_driver_style_steps20.json
  timeout 600 python bench.py --no-cpu-baseline --no-extra --precision fp32 > $O/bench_fp32_mfma.log 2>&1 < /dev/null
  grep '"metric"' $O/bench_fp32_mfma.log | tail -1 > $O/bench_fp32_mfma.json
  timeout 600 python bench.py --no-cpu-baseline --precision bf16_storage > $O/bench_bf16_storage.log 2>&1 < /dev/null
  grep '"metric"' $O/bench_bf16_storage.log | tail -1 > $O/bench_bf16_storage.json
  rm -f $O/bench_*.log
fi
if want prof; then
  for w in seg reg joint joint_smooth; do
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$w -- python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/prof_$w.log 2>&1 < /dev/null
    f=$(ls $O/prof_$w/*/*.db 2>/dev/null | head -1)
    if [ -n "$f" ]; then
      python tools/rocpd_summary.py "$f" --top 60 > $O/${w}_kernel_stats.txt 2>&1 < /dev/null
      python tools/rocpd_timeline.py "$f" $([ ${w%%_*} = joint ] && echo --adam-per-step 2) > $O/${w}_timeline.txt 2>&1 < /dev/null
    fi
    rm -rf $O/prof_$w $O/prof_$w.log
    timeout 600 python tools/step_calls.py $w 2>&1 | grep -v amdgpu.ids > $O/${w}_calls.txt
  done
fi
if want traffic; then
  for w in seg reg joint; do
    bash tools/pmc_step.sh $w 3 > /dev/null 2>&1
    python tools/pmc_step_summary.py gpurun_out/pmc_step $w --out $O/step_traffic_$w > /dev/null 2>&1
  done
fi
if want kernels; then
  timeout 600 python tools/bench_losses.py 2>&1 | grep -v amdgpu.ids | grep -v '^\[' > $O/hbm_bound_calls.txt
  timeout 300 python tools/bench_warp.py 2>&1 | grep -v amdgpu.ids > $O/gather_kernels.txt
  LAYERS="32,16,16,2,160,192,160 16,0,16,2,160,192,160 8,0,16,2,160,192,160 64,32,32,2,80,96,80 32,0,32,2,80,96,80 64,64,64,2,40,48,40"
  DA_MATRIX_MODE=2 timeout 300 python tools/bench_pointwise.py 2>&1 | grep -v amdgpu.ids > $O/pointwise_calls.txt
  { echo "# a tiny kernel on the main stream beside a whole-GPU kernel on a side stream (tools/debug/concurrency_probe.py): probe ms / big-kernel ms / probe start after the big kernel's start";
    for a in "--layer 8,16,3,1,160,192,160 --what wgrad" "--layer 32,16,16,1,160,192,160 --what wgrad" "--layer 32,16,16,1,160,192,160 --what fwd"; do DA_MATRIX_MODE=2 timeout 300 python tools/debug/concurrency_probe.py $a 2>&1 | grep -v amdgpu.ids; done; } > $O/concurrency_probe.txt
  if ls deepatlas_amd/csrc/libda_D1.so > /dev/null 2>&1; then   # (needs the variant libraries: tools/ab/build_variant.sh D$v pointwise_mfma.hip "-DDA_DB_ABL=$v" for v in 1 2 4 8 15)
  { echo "# the fused up-sampler backward (deconv_bn_bwd_kernel) with pieces removed (tools/ab/deconv_bwd_ablate.sh; timing only): DA_DB_ABL bits 1 no MFMAs, 2 no weight-gradient half, 4 no gout / y loads, 8 no cross-wave sum + dx store";
    if ls deepatlas_amd/csrc/libda_D1.so > /dev/null 2>&1; then bash tools/ab/deconv_bwd_ablate.sh 2>&1 | grep -v "simple_timer\|amdgpu.ids"; fi; } > $O/deconv_bwd_ablate.txt
  fi
  echo "# DA_MATRIX_MODE=2 (fp32_split: two-term fp16 split), tools/bench_conv.py --layer C1,C2,Cout,N,D,H,W; one process per layer and variant, same box" > $O/conv_layers_isolated.txt
  for e in "" "DA_FWDSP=1" "DA_FWDSP=1 DA_FWDSP8=1"; do
    echo "== ${e:-shipped kernels (conv3d_mfma.hip)}" >> $O/conv_layers_isolated.txt
    for L in $LAYERS; do env DA_MATRIX_MODE=2 $e timeout 600 python tools/bench_conv.py --layer $L --what fwd,fwdstats,fwdpro,dgrad,wgrad 2>&1 | grep -v amdgpu.ids >> $O/conv_layers_isolated.txt; done
  done
  echo "# the same instruction streams on ALL-ZERO operands (DA_ZERO=1: no toggling in the multipliers): what the power limit costs" > $O/conv_zero_operands.txt
  for e in "" "DA_ZERO=1"; do
    echo "== ${e:-random operands}" >> $O/conv_zero_operands.txt
    for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160; do env DA_MATRIX_MODE=2 $e timeout 600 python tools/bench_conv.py --layer $L --what fwdstats,dgrad,wgrad 2>&1 | grep -v amdgpu.ids >> $O/conv_zero_operands.txt; done
  done
  DA_MATRIX_MODE=2 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_conv -- python tools/bench_conv.py --layer 32,16,16,2,160,192,160 --iters 5 > $O/prof_conv.log 2>&1 < /dev/null
  f=$(ls $O/prof_conv/*/*.db 2>/dev/null | head -1)
  if [ -n "$f" ]; then python tools/rocpd_summary.py "$f" > $O/conv3d_48to16_kernel_stats.txt 2>&1 < /dev/null; fi
  rm -rf $O/prof_conv $O/prof_conv.log
fi
if want ablate; then
  # in-step ablation of the forward / data-gradient kernel (the opt-in form carries the switches): pieces removed one at a time, timing only
  L=$PWD/deepatlas_amd/csrc
  { echo "# seg step (batch 2, 160x192x160) with pieces of the split forward / data-gradient kernel removed (DA_FWDSP=1; wrong results, timing only)";
    echo "# DA_ABLATE bits: 1 no staging loads, 2 no epilogue, 4 no conversion + hand-over barriers, 32 no output stores, 64 no statistics; libda_A8 / A16: no MFMAs / no activation fragment reads";
    A8=""; A16=""; A24=""
    [ -f $L/libda_A8.so ] && A8="DA_FWDSP=1 DA_LIB=$L/libda_A8.so" && A16="DA_FWDSP=1 DA_LIB=$L/libda_A16.so" && A24="DA_FWDSP=1 DA_LIB=$L/libda_A24.so"
    bash tools/ab/ablate_step.sh seg "DA_FWDSP=1" "DA_FWDSP=1 DA_ABLATE=1" "DA_FWDSP=1 DA_ABLATE=2" "DA_FWDSP=1 DA_ABLATE=4" "DA_FWDSP=1 DA_ABLATE=32" "DA_FWDSP=1 DA_ABLATE=64" "DA_FWDSP=1 DA_ABLATE=7" ${A8:+"$A8"} ${A16:+"$A16"} ${A24:+"$A24"} 2>&1; } > $O/ablate_step_seg.txt
fi
if want pmc; then
  bash tools/pmc_conv.sh 2 > $O/pmc.log 2>&1
fi
ls -la $O
