cd $GRAFT_REPO_ROOT
for L in 32,16,16,2,160,192,160 16,0,16,2,160,192,160 64,32,32,2,80,96,80; do
 for lib in "" tools/ab/libda_fakesplit.so; do
  echo "== layer $L lib=${lib:-shipped}"
  DA_LIB=$lib DA_MATRIX_MODE=2 timeout 300 python tools/bench_conv.py --layer $L --what fwdstats,dgrad,wgrad 2>&1 | grep -v amdgpu.ids
 done
done
