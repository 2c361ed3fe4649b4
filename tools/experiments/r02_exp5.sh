# round 2, experiment 5: full GPU suite (context API, CLI, bench rehearsal), host-entry timing, default bench, stamps
set -o pipefail
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity_at_scale_suite.jsonl
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/r02_exp5_pytest.log 2>&1; rc=$?
tail -22 gpurun_out/r02_exp5_pytest.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
for s in "lbfgsb f64" "lm f32" "lm f64" "loglin f64 gaussian"; do python tools/host_entry_bench.py $s 2>/dev/null; done | tee gpurun_out/r02_exp5_host_entry.jsonl
T2FIT_COPY_THREADS=4 python tools/host_entry_bench.py lbfgsb f64 2>/dev/null | tee -a gpurun_out/r02_exp5_host_entry.jsonl
T2FIT_COPY_THREADS=16 python tools/host_entry_bench.py lbfgsb f64 2>/dev/null | tee -a gpurun_out/r02_exp5_host_entry.jsonl
python bench.py > gpurun_out/r02_exp5_bench.json 2> gpurun_out/r02_exp5_bench.err; cat gpurun_out/r02_exp5_bench.json
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_stamps.so python bench.py --no-also --cpu-seconds 0 --steps 2 --warmup 1 > gpurun_out/r02_exp5_stamps.json 2> gpurun_out/r02_exp5_stamps.err
grep "t2fit blocks" gpurun_out/r02_exp5_stamps.err | tail -11
