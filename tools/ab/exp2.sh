cd $GRAFT_REPO_ROOT
L=32,16,16,2,160,192,160
for ab in 0 1 2 4 5 7; do
  echo "== bf16 storage, layer $L, DA_ABLATE=$ab (1 no staging loads, 2 no epilogue, 4 no LDS writes + barriers)"
  DA_ABLATE=$ab DA_MATRIX_MODE=1 timeout 300 python tools/bench_conv.py --layer $L --what fwd,fwdstats,dgrad --bf16-storage 2>&1 | grep -v amdgpu.ids
done
echo "== fp32 storage bf16 matrix mode"
DA_MATRIX_MODE=1 timeout 300 python tools/bench_conv.py --layer $L --what fwd,fwdstats,dgrad,wgrad 2>&1 | grep -v amdgpu.ids
DA_MATRIX_MODE=1 timeout 300 python tools/bench_conv.py --layer $L --what wgrad --bf16-storage 2>&1 | grep -v amdgpu.ids
for L2 in 16,0,16,2,160,192,160 64,32,32,2,80,96,80; do DA_MATRIX_MODE=1 timeout 300 python tools/bench_conv.py --layer $L2 --what fwdstats,dgrad,wgrad --bf16-storage 2>&1 | grep -v amdgpu.ids; done
