# round 3: shared i0e loop with two register sets for the coefficient chunks taken in turn (two chunks per trip, no copies
# between the vector loads and their use).  P = tools/diag/libt2fit_p.so (exp25 form), B = in-tree.
cd $GRAFT_REPO_ROOT
D=$PWD/tools/diag
run() { T2FIT_LIB=$D/libt2fit_p.so python tools/kernel_ab.py P "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py B "$@" 2>/dev/null | tail -1; }
run --fit rician --shape 180 256 256 --nte 6 && run --fit rician && run --fit rician --shape 64 256 256 --nte 3 && run --fit rician --shape 64 256 256 --nte 7 --extras && run --fit rician --shape 180 256 256 --nte 6 &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rician or large_volume_kernels or stable_set or bad_samples" 2>&1 | tail -3
