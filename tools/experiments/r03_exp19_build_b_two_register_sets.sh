# round 3: build_b() with two register sets for the pairs taken in turn (the fetch of pair p + 1 lands where it is read: five
# 64-bit moves and two waits less per pair).  A = tools/diag/libt2fit_base.so (before), B = in-tree.  Digests must be equal.
cd $GRAFT_REPO_ROOT
A=$PWD/tools/diag/libt2fit_base.so
run() { T2FIT_LIB=$A python tools/kernel_ab.py A "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py B "$@" 2>/dev/null | tail -1; }
run && run --no_prior && run --shape 180 256 256 --nte 6 && run --fit gaussian --shape 180 256 256 --nte 6 && run --fit gaussian --no_prior &&
run --fit rician --shape 180 256 256 --nte 6 && run --shape 64 256 256 --nte 7 --extras && run --shape 8 256 256 --nte 9 && run --fit gaussian --shape 8 256 256 --nte 9 && run &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_difference or large_volume_kernels or stable_set or echo_train or bad_samples" 2>&1 | tail -3
