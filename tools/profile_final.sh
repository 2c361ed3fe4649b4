# usage (GPU box, repo root): bash tools/profile_final.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of `python bench.py --cpu-seconds 0` (no CPU-baseline worker pool forked under the profiler); 2. PMC passes (own runs, never combined with a trace)
# for the three fit kernels; 3. the other BASELINE.json configurations; 4. config-5 streaming.
R=$GRAFT_REPO_ROOT; TAG=$1; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py --cpu-seconds 0 > $R/gpurun_out/prof_${TAG}.json 2> $R/gpurun_out/prof_${TAG}.err
cd $R
for spec in "lbfgsb:--solver lbfgsb --no-also" "lmf32:--solver lm --precision f32" "loglin:--solver loglin --fit gaussian"; do
  t=${spec%%:*}; args=${spec#*:}
  bash tools/pmc_passes.sh "$args" $t > /dev/null 2>&1
  bash tools/pmc_extra.sh "$args" $t > /dev/null 2>&1
done
python tools/pmc_summary.py lbfgsb persistent > gpurun_out/pmc_${TAG}_lbfgsb.txt
python tools/pmc_summary.py lmf32 persistent > gpurun_out/pmc_${TAG}_lmf32.txt
python tools/pmc_summary.py loglin loglin > gpurun_out/pmc_${TAG}_loglin.txt
python tools/pmc_summary.py lbfgsb residuals > gpurun_out/pmc_${TAG}_residuals.txt
bash tools/bench_configs.sh > gpurun_out/bench_configs_${TAG}.jsonl 2>&1
python tools/stream_bench.py 16 lm f32 > gpurun_out/stream_${TAG}_lm_f32.json 2>/dev/null
python tools/stream_bench.py 16 lbfgsb f64 > gpurun_out/stream_${TAG}_lbfgsb_f64.json 2>/dev/null
python tools/stream_bench.py 16 loglin f64 gaussian > gpurun_out/stream_${TAG}_loglin.json 2>/dev/null
cat gpurun_out/prof_${TAG}.json
