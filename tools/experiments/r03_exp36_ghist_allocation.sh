R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export T2FIT_LIB=$R/tools/diag/libt2fit_gstatic.so
for v in async static; do
  if [ $v = static ]; then export T2FIT_GHIST_STATIC=1; fi
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_gs_${v}_w -- python3 $R/bench.py --no-also --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_gs_${v}_w.log 2>&1
done
python3 - <<PY
import csv, glob
for v in ("async", "static"):
    for f in glob.glob("$R/gpurun_out/pmc_gs_%s_w/*/*counter_collection.csv" % v):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "fit_persistent" in r["Kernel_Name"]]
        if vals: print(v, "WRITE_SIZE %.1f MiB per launch (mean of %d)" % (sum(vals) / len(vals) / 1024.0, len(vals)))
PY
T2FIT_GHIST_STATIC=1 python3 $R/tools/kernel_ab.py static 2>/dev/null | tail -1
unset T2FIT_GHIST_STATIC; python3 $R/tools/kernel_ab.py async 2>/dev/null | tail -1
