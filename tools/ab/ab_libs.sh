#!/bin/bash
# per-layer timings under several alternative builds: tools/ab/ab_libs.sh "NAME1 NAME2" ["layers"] [what]   (libda_NAME.so from tools/ab/build_variant.sh; NAME "shipped" = the in-tree library)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
LAYERS=${2:-"32,16,16,2,160,192,160 16,0,16,2,160,192,160 8,0,16,2,160,192,160 64,32,32,2,80,96,80"}
WHAT=${3:-fwd,fwdstats,fwdpro,dgrad}
for n in $1; do echo "== lib $n"; lib=$PWD/deepatlas_amd/csrc/libda_$n.so; [ "$n" = shipped ] && lib=$PWD/deepatlas_amd/csrc/libdeepatlas_hip.so
  for L in $LAYERS; do env DA_MATRIX_MODE=2 DA_LIB=$lib $EXTRA_ENV python tools/bench_conv.py --layer $L --what $WHAT 2>&1 | grep -v amdgpu.ids; done; done
