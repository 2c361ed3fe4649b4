# round 3: tools/diag/l2_write_probe under rocprofv3 --pmc WRITE_SIZE (own pass, no trace): do repeated stores to a 10 MiB
# L2-resident buffer reach the fabric every time?
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_l2probe_w -- $R/tools/diag/l2_write_probe > $R/gpurun_out/l2probe.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_l2probe_f -- $R/tools/diag/l2_write_probe >> $R/gpurun_out/l2probe.log 2>&1
grep "buffer\|partial stores" $R/gpurun_out/l2probe.log | head -5
python3 - <<PY
import csv, glob
for tag in ("w", "f"):
    for f in glob.glob("$R/gpurun_out/pmc_l2probe_%s/*/*counter_collection.csv" % tag):
        for r in csv.DictReader(open(f)):
            if "rewrite" in r["Kernel_Name"]:
                print(tag, r["Kernel_Name"][:40], r["Counter_Name"], "%.1f MiB" % (float(r["Counter_Value"]) / 1024.0))
PY
