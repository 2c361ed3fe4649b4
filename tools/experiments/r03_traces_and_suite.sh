R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03b -- python3 $R/bench.py --cpu-seconds 0 --no-also > $R/gpurun_out/prof_r03b.json 2> $R/gpurun_out/prof_r03b.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03b_all -- python3 $R/bench.py --cpu-seconds 0 > $R/gpurun_out/prof_r03b_all.json 2> $R/gpurun_out/prof_r03b_all.err &&
cd $R && timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest4.log 2>&1; tail -3 gpurun_out/r03_gputest4.log; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8
