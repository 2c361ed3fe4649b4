# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# kernel-trace stats of the default bench command (python bench.py, no flags), then PMC passes (own runs) for HBM traffic and SQ counters
R=$GRAFT_REPO_ROOT; TAG=$1; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/bench.py --cpu-seconds 0 > $R/gpurun_out/prof_${TAG}.json 2> $R/gpurun_out/prof_${TAG}.err
cat $R/gpurun_out/prof_${TAG}.json
cd $R; bash tools/pmc_passes.sh "--solver lbfgsb --no-also" lbfgsb > /dev/null 2>&1; bash tools/pmc_passes.sh "--solver lm --precision f32" lmf32 > /dev/null 2>&1
python tools/pmc_summary.py lbfgsb persistent > gpurun_out/pmc_${TAG}_lbfgsb.txt; python tools/pmc_summary.py lmf32 persistent > gpurun_out/pmc_${TAG}_lmf32.txt
python tools/pmc_summary.py lbfgsb residuals > gpurun_out/pmc_${TAG}_residuals.txt
cat gpurun_out/pmc_${TAG}_lbfgsb.txt gpurun_out/pmc_${TAG}_lmf32.txt gpurun_out/pmc_${TAG}_residuals.txt | grep -E "FETCH|WRITE"
