cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in "" D1 D2 D4 D8 D15; do  # (variants: tools/ab/build_variant.sh D$v pointwise_mfma.hip "-DDA_DB_ABL=$v")
  lib=""; [ -n "$L" ] && lib=deepatlas_amd/csrc/libda_$L.so
  rm -rf gpurun_out/dbp; DA_LIB=$lib rocprofv3 --kernel-trace --stats -d gpurun_out/dbp -o x --output-format csv -- python tools/bench_pointwise.py --only "bn bwd" > gpurun_out/dbp.log 2>&1
  echo "== ${L:-shipped}: $(grep 'bn bwd' gpurun_out/dbp.log)"
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/dbp/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["TotalDurationNs"])>2e5: print("   %-60s %4s %9.1f us"%(r["Name"].replace("(anonymous namespace)::","")[:60],r["Calls"],float(r["AverageNs"])/1e3))
PY
done
