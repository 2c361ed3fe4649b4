# round 3: tools/diag/l2_write_probe says the L2 keeps a rewritten buffer (31 of 412 MiB reach the fabric) until a read stream passes
# through it (400 MiB), and keeps it again when that stream is read with the non-temporal hint (47 MiB).  So: the fit kernels'
# sample loads non-temporal (tools/diag/libt2fit_ntload.so) -- the ring's global part should then stay in L2.  WRITE_SIZE / FETCH_SIZE and time.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in plain nt; do
  if [ $v = nt ]; then export T2FIT_LIB=$R/tools/diag/libt2fit_ntload.so; fi
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_nt_${v}_w -- python3 $R/bench.py --no-also --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_nt_${v}_w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_nt_${v}_f -- python3 $R/bench.py --no-also --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_nt_${v}_f.log 2>&1
done
unset T2FIT_LIB
python3 - <<PY
import csv, glob
for v in ("plain", "nt"):
    for tag in ("w", "f"):
        for f in glob.glob("$R/gpurun_out/pmc_nt_%s_%s/*/*counter_collection.csv" % (v, tag)):
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "fit_persistent" in r["Kernel_Name"]]
            if vals: print(v, "WRITE_SIZE" if tag == "w" else "FETCH_SIZE", "%.1f MiB per launch" % (sum(vals) / len(vals) / 1024.0))
PY
cd $R
run() { python tools/kernel_ab.py plain "$@" 2>/dev/null | tail -1 && T2FIT_LIB=$R/tools/diag/libt2fit_ntload.so python tools/kernel_ab.py nt "$@" 2>/dev/null | tail -1; }
run && run --shape 180 256 256 --nte 6 && run --fit gaussian --shape 180 256 256 --nte 6 && run --fit rician --shape 180 256 256 --nte 6 && run --solver lm --precision f32 && run --shape 32 256 256 && run
