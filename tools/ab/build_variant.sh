#!/bin/bash
# Build an alternative libdeepatlas_hip with one source recompiled under extra -D flags (A/B on the GPU box through DA_LIB).
#   tools/ab/build_variant.sh NAME SOURCE.hip "-DFOO=1 -DBAR=2"   ->  deepatlas_amd/csrc/libda_NAME.so
# Every other object is taken from the current in-tree build (run __graft_entry__.build() first).
set -e
cd "$(dirname "$0")/../../deepatlas_amd/csrc"
name=$1; src=$2; flags=$3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $src -o /tmp/var_${name}.o 2> /tmp/var_${name}.err || { grep -i -A3 "error" /tmp/var_${name}.err; exit 1; }
objs=$(ls *.o | grep -v "^${src%.hip}.o$" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/var_${name}.o -o libda_${name}.so
if [ "$4" = "regs" ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags -S --cuda-device-only $src -o /tmp/var_${name}.s 2>/dev/null; python ../../tools/kernel_regs.py /tmp/var_${name}.s "${5:-kernel}" | c++filt | sed 's/(anonymous namespace):://g' | cut -c1-160; echo "scratch ops: $(grep -c scratch_ /tmp/var_${name}.s)"; fi
echo "built libda_${name}.so"
