#!/bin/bash
# GPU box (via gpurun): HBM-side traffic + SQ counters of the dominant conv layer (48 -> 16 = concat 32 + 16, batch 2, 160x192x160).
# One PROCESS per op (fwdstats | dgrad | wgrad), so every non-torch kernel row of a CSV belongs to that op whatever the kernels are called, and one
# counter BLOCK per rocprofv3 pass as MI355X_MICROARCH.md prescribes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2):
#   FETCH_SIZE | WRITE_SIZE | request-size mix (RDREQ, _32B, _64B, _128B) | size-weighted DRAM / GMI / IO 32-byte units | SQ block
# The same TCC passes run over tools/ubench/fetch_calib.hip (known byte counts in the kernels' own request shapes): tools/pmc_summary.py
# uses them to decide which counter reproduces a known byte count and to correct FETCH_SIZE with the kernel's OWN request mix.
# usage: tools/pmc_conv.sh [matrix mode: 2 = fp32_split (default, the headline), 0 = fp32 MFMA] [ops, default "fwdstats dgrad wgrad"]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export DA_MATRIX_MODE=${1:-2}
OPS=${2:-"fwdstats dgrad wgrad"}
L=32,16,16,2,160,192,160
O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
echo $DA_MATRIX_MODE > $O/matrix_mode.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/ubench/fetch_calib.hip -o /tmp/fetch_calib > $O/calib_build.log 2>&1
pass() {   # pass NAME "COUNTERS"
  for op in $OPS; do
    timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/tmp -- python tools/bench_conv.py --layer $L --what $op --iters 3 > $O/$1_$op.log 2>&1 < /dev/null
    f=$(ls $O/tmp/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then cp "$f" $O/conv3d_48to16_${op}_$1.csv; rm -f $O/$1_$op.log; else echo "pass $1 op $op: no counter csv"; tail -5 $O/$1_$op.log; fi
    rm -rf $O/tmp
  done
  if [ "$1" != SQ ]; then
    timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/tmp -- /tmp/fetch_calib > $O/$1_calib.log 2>&1 < /dev/null
    f=$(ls $O/tmp/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then cp "$f" $O/fetch_calib_$1.csv; rm -f $O/$1_calib.log; else echo "pass $1 calibration: no counter csv"; tail -5 $O/$1_calib.log; fi
    rm -rf $O/tmp
  fi
}
pass FETCH_SIZE "FETCH_SIZE"
pass WRITE_SIZE "WRITE_SIZE"
pass RDMIX "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
pass RDW32 "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"
pass L1L2 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
pass SQ "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
git rev-parse --short HEAD > $O/commit.txt 2>/dev/null || true
ls -la $O
