#!/bin/bash
# One parameterised A/B driver for the GPU box (replaces the per-experiment exp_*.sh scripts of rounds 3 - 4; their outputs are under profiles/).
#   tools/ab/ab.sh layers  [-w fwd,fwdstats,dgrad,wgrad] [-l "L1 L2 ..."] [-e "VAR=1 VAR2=0" -e "..."]   per-layer timings (tools/bench_conv.py), once per -e setting
#   tools/ab/ab.sh step    [-W seg|reg|joint] [-s steps] [-r reps] [-e "VAR=1" -e "..."]                  ms_per_step of bench.py, default + once per -e setting
#   tools/ab/ab.sh sweep   VAR "v0 v1 v2 ..." [-w ...] [-l ...]                                           one environment variable swept over values (DA_ABLATE, DA_WG_ABLATE, ...)
#   tools/ab/ab.sh pmc     KERNEL_SUBSTR "COUNTERS ..." [-w wgrad] [-l ...]                               one rocprofv3 --pmc pass, per-launch averages of kernels matching the substring
#   tools/ab/ab.sh lib     PATH.so ...                                                                    layers + step with DA_LIB pointing at an alternative build of the library
# A layer is C1,C2,Cout,N,D,H,W.  Everything runs in the shipped split matrix mode unless -e overrides DA_MATRIX_MODE.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mode=$1; shift
LAYERS="32,16,16,2,160,192,160 16,0,16,2,160,192,160 8,0,16,2,160,192,160 64,32,32,2,80,96,80 32,0,32,2,80,96,80 64,64,64,2,40,48,40"
WHAT=fwd,fwdstats,dgrad,wgrad; WORK=seg; STEPS=10; REPS=2; ENVS=(); POS=()
while [ $# -gt 0 ]; do case $1 in
  -w) WHAT=$2; shift 2;; -l) LAYERS=$2; shift 2;; -e) ENVS+=("$2"); shift 2;; -W) WORK=$2; shift 2;; -s) STEPS=$2; shift 2;; -r) REPS=$2; shift 2;;
  *) POS+=("$1"); shift;; esac; done
layers() { for L in $LAYERS; do env DA_MATRIX_MODE=2 $1 timeout 300 python tools/bench_conv.py --layer $L --what $WHAT 2>&1 | grep -v amdgpu.ids; done; }
step() { env $1 timeout 900 python bench.py --workload $WORK --steps $STEPS --warmup 3 --no-cpu-baseline --no-extra --no-profile 2>&1 | grep -v amdgpu.ids | grep -o '"ms_per_step": [0-9.]*' | head -1; }
case $mode in
  layers) [ ${#ENVS[@]} -eq 0 ] && ENVS=(""); for e in "${ENVS[@]}"; do echo "== env: ${e:-default}"; layers "$e"; done;;
  step)   for e in "" "${ENVS[@]}"; do for rep in $(seq $REPS); do echo "== $WORK env: ${e:-default}"; step "$e"; done; done;;
  sweep)  for v in ${POS[1]}; do echo "== ${POS[0]}=$v"; layers "${POS[0]}=$v"; done;;
  lib)    for lib in "" "${POS[@]}"; do echo "== lib: ${lib:-shipped}"; layers "DA_LIB=$lib"; step "DA_LIB=$lib"; done;;
  pmc)    O=gpurun_out/abpmc; rm -rf $O; mkdir -p $O
          for L in $LAYERS; do
            timeout 300 rocprofv3 --pmc ${POS[1]} --kernel-trace --output-format csv -d $O/p -- env DA_MATRIX_MODE=2 python tools/bench_conv.py --layer $L --what $WHAT --iters 2 > $O/log.txt 2>&1 < /dev/null
            f=$(ls $O/p/*/*counter_collection.csv 2>/dev/null | head -1)
            if [ -n "$f" ]; then python tools/pmc_summary.py --averages "$f" "${POS[0]}" "$L"; else echo "$L: no csv"; tail -3 $O/log.txt; fi
            rm -rf $O/p
          done;;
  *) sed -n 2,9p "$0";;
esac
