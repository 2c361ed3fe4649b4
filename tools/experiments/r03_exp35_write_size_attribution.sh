# round 3: where do the fit kernel's 1.2 GB of fabric writes come from?  WRITE_SIZE / FETCH_SIZE (own passes) of the headline launch
# with the split ring (in-tree: one number of each pair in global memory) and with the whole ring in LDS (tools/diag/libt2fit_nosplit.so,
# six waves per CU), and tools/diag/l2_write_probe: a 10 MiB buffer rewritten 200 times (the L2 absorbs 96 % of those stores).
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in split nosplit; do
  if [ $v = nosplit ]; then export T2FIT_LIB=$R/tools/diag/libt2fit_nosplit.so; fi
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_ws_${v}_w -- python3 $R/bench.py --no-also --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_ws_${v}_w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_ws_${v}_f -- python3 $R/bench.py --no-also --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_ws_${v}_f.log 2>&1
done
python3 - <<PY
import csv, glob
for v in ("split", "nosplit"):
    for tag in ("w", "f"):
        for f in glob.glob("$R/gpurun_out/pmc_ws_%s_%s/*/*counter_collection.csv" % (v, tag)):
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "fit_persistent" in r["Kernel_Name"]]
            if vals: print(v, "WRITE_SIZE" if tag == "w" else "FETCH_SIZE", "%.1f MiB per launch (mean of %d)" % (sum(vals) / len(vals) / 1024.0, len(vals)))
PY
