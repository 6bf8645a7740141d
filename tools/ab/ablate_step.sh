#!/bin/bash
# What is the training step sensitive to?  ms_per_step of a workload with pieces of the forward / data-gradient kernel removed (wrong results; timing only).
#   tools/ab/ablate_step.sh seg "DA_ABLATE=1" "DA_ABLATE=2" "DA_LIB=.../libda_X.so" ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
W=${1:-seg}; shift
run() { env $1 timeout 600 python bench.py --workload $W --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-profile 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for e in "X=0" "$@" "X=1"; do echo "== $e: $(run "$e")"; done
