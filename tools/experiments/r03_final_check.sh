cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/final4_gpu_suite.log 2>&1 && python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final4_smoke.log 2>&1 && python bench.py > gpurun_out/final4_bench.json 2> gpurun_out/final4_bench.err
tail -2 gpurun_out/final4_gpu_suite.log; tail -1 gpurun_out/final4_smoke.log
