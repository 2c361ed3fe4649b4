# usage (GPU box, repo root): bash tools/profile_r03.sh <tag>     -- the round's evidence in one call
# 1. rocprofv3 --kernel-trace --stats of `python bench.py --cpu-seconds 0`; 2. PMC passes (own runs, never combined with a
# trace) for the fit kernels; 3. the other BASELINE.json configurations; 4. config-5 streaming; 5. host entry; 6. overlap checks
R=$GRAFT_REPO_ROOT; TAG=$1; cd /tmp; export TMPDIR=/tmp
# (--no-also: the headline kernel's symbol then serves the headline configuration only -- `also_noprior` runs the same
# instantiation -- so the kernel's average in the stats is the figure `roofline.kernel_ms` must agree with)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py --cpu-seconds 0 --no-also > $R/gpurun_out/prof_${TAG}.json 2> $R/gpurun_out/prof_${TAG}.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_all -- python3 $R/bench.py --cpu-seconds 0 > $R/gpurun_out/prof_${TAG}_all.json 2> $R/gpurun_out/prof_${TAG}_all.err
cd $R
echo "kernel trace done"
for spec in "lbfgsb:--solver lbfgsb --no-also" "rician:--fit rician --shape 180 256 256 --n-te 6 --no-also" "lmf32:--solver lm --precision f32" "lmf64:--solver lm --precision f64" "loglin:--solver loglin --fit gaussian"; do
  t=${spec%%:*}; args=${spec#*:}
  bash tools/pmc_passes.sh "$args" $t > /dev/null 2>&1
  bash tools/pmc_extra.sh "$args" $t > /dev/null 2>&1
  echo "pmc $t done"
done
python tools/pmc_summary.py lbfgsb persistent > gpurun_out/pmc_${TAG}_lbfgsb.txt
python tools/pmc_summary.py rician persistent > gpurun_out/pmc_${TAG}_rician.txt
python tools/pmc_summary.py lmf32 persistent > gpurun_out/pmc_${TAG}_lmf32.txt
python tools/pmc_summary.py lmf64 persistent > gpurun_out/pmc_${TAG}_lmf64.txt
python tools/pmc_summary.py loglin loglin > gpurun_out/pmc_${TAG}_loglin.txt
python tools/pmc_summary.py lbfgsb residuals > gpurun_out/pmc_${TAG}_residuals.txt
bash tools/bench_configs.sh > gpurun_out/bench_configs_${TAG}.jsonl 2>&1
echo "configs done"
python tools/stream_bench.py 16 lm f32 > gpurun_out/stream_${TAG}_lm_f32.json 2>/dev/null
python tools/stream_bench.py 16 lbfgsb f64 > gpurun_out/stream_${TAG}_lbfgsb_f64.json 2>/dev/null
python tools/stream_bench.py 16 loglin f64 gaussian > gpurun_out/stream_${TAG}_loglin.json 2>/dev/null
echo "stream done"
for s in "lbfgsb f64" "lm f32" "lm f64" "loglin f64 gaussian"; do python tools/host_entry_bench.py $s 2>/dev/null; done > gpurun_out/host_entry_${TAG}.jsonl
python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
cat gpurun_out/bench_${TAG}.json
