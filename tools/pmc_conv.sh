#!/bin/bash
# GPU box (via gpurun): HBM traffic of the dominant conv layer (48->16 = concat 32+16, batch 2, 160x192x160), one counter per pass
# as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Output: gpurun_out/pmc/*.csv
# usage: tools/pmc_conv.sh [matrix mode: 2 = fp32_split (default, the headline), 0 = fp32 MFMA]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export DA_MATRIX_MODE=${1:-2}
O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
echo $DA_MATRIX_MODE > $O/matrix_mode.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python tools/bench_conv.py --layer 32,16,16,2,160,192,160 --what fwd,fwdstats,wgrad --iters 3 > $O/$c.log 2>&1 < /dev/null
  f=$(ls $O/$c/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" $O/conv3d_48to16_$c.csv; fi
  rm -rf $O/$c
  # the data gradient in its own process: in split mode it is two launches, one of them the forward's kernel at the forward's grid
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python tools/bench_conv.py --layer 32,16,16,2,160,192,160 --what dgrad --iters 3 > $O/${c}_dgrad.log 2>&1 < /dev/null
  f=$(ls $O/$c/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" $O/conv3d_48to16_dgrad_$c.csv; fi
  rm -rf $O/$c
done
# third pass: matrix-pipe utilisation of the same kernels (SQ block: 8 slots, GRBM: 2 -- no TCC counters in this pass)
c=SQ
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/$c -- python tools/bench_conv.py --layer 32,16,16,2,160,192,160 --what fwd,fwdstats,wgrad --iters 3 > $O/$c.log 2>&1 < /dev/null
f=$(ls $O/$c/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/conv3d_48to16_$c.csv; fi
rm -rf $O/$c
# fourth pass: FETCH_SIZE calibration on known byte counts in the conv kernels' own request shapes (tools/ubench/fetch_calib.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/ubench/fetch_calib.hip -o /tmp/fetch_calib > $O/calib_build.log 2>&1
c=CALIB
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$c -- /tmp/fetch_calib > $O/$c.log 2>&1 < /dev/null
f=$(ls $O/$c/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $O/fetch_calib_FETCH_SIZE.csv; fi
rm -rf $O/$c
git rev-parse --short HEAD > $O/commit.txt 2>/dev/null || true
ls -la $O
