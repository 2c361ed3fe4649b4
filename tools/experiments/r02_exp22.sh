cd $GRAFT_REPO_ROOT
python tools/kernel_ms.py default
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_stamps.so python bench.py --no-also --cpu-seconds 0 --steps 2 --warmup 1 > gpurun_out/r02_exp22_stamps.json 2> gpurun_out/r02_exp22_stamps.err
grep "t2fit blocks" gpurun_out/r02_exp22_stamps.err | tail -11
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stable_set or matches_reference or size_independent or host_entry or echo_train or 256cubed" 2>&1 | tail -3
