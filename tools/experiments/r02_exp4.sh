# round 2, experiment 4: the whole GPU suite with the new parity tests (stable sets, 20k-voxel parity, configs at size)
set -o pipefail
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity_at_scale_suite.jsonl
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r02_exp4_pytest.log 2>&1; rc=$?
tail -45 gpurun_out/r02_exp4_pytest.log
cat gpurun_out/parity_at_scale_suite.jsonl 2>/dev/null
exit $rc
