#!/bin/bash
# A/B library for the split-arithmetic ablation: the same sources, da_split3 without arithmetic (-DDA_FAKE_SPLIT; wrong results, same data volume).
# Run after __graft_entry__.build(); tools/ab/exp1.sh compares it with the shipped library through tools/bench_conv.py (DA_LIB).
cd "$(dirname "$0")/../../deepatlas_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDA_FAKE_SPLIT -c conv3d_mfma.hip -o /tmp/conv3d_mfma_fake.o || exit 1
ls *.o | grep -v '^conv3d_mfma.o$' | xargs /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/conv3d_mfma_fake.o -o ../../tools/ab/libda_fakesplit.so
