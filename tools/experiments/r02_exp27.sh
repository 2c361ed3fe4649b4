cd $GRAFT_REPO_ROOT
bash tools/sweep.sh T2FIT_REFILL_MIN "4 6 8 12" "--solver lbfgsb" 2>&1 | tee gpurun_out/r02_exp27_refill.txt
bash tools/sweep.sh T2FIT_PERSISTENT_BLOCKS "256 512 2048" "--solver lbfgsb" 2>&1 | tee -a gpurun_out/r02_exp27_refill.txt
rm -f gpurun_out/parity_at_scale_suite.jsonl
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r02_exp27_pytest.log 2>&1; tail -4 gpurun_out/r02_exp27_pytest.log
