# round 3: Rician lane, shared 30-step i0e loop: 32 / x and the division by sqrt(x) of the (8, inf) series from ONE reciprocal
# square root per echo (t2_i0e4_by_lane near = true) + the base square root of the least-squares evaluation without its seed cap.
# A = tools/diag/libt2fit_base.so (before both), B = in-tree.  Digests must be equal.
cd $GRAFT_REPO_ROOT
A=$PWD/tools/diag/libt2fit_base.so
run() { T2FIT_LIB=$A python tools/kernel_ab.py A "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py B "$@" 2>/dev/null | tail -1; }
run --fit rician --shape 180 256 256 --nte 6 && run --fit rician && run --fit rician --legacy --shape 180 256 256 --nte 6 &&
run --fit rician --shape 64 256 256 --nte 3 && run --fit rician --shape 64 256 256 --nte 7 --extras && run --fit rician --shape 64 256 256 --nte 5 --no_prior &&
run --fit rician --shape 8 256 256 --nte 8 && run --fit rician --shape 8 256 256 --nte 9 && run && run --shape 180 256 256 --nte 6 && run --fit rician --shape 180 256 256 --nte 6 &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_difference or large_volume_kernels or stable_set or echo_train or bad_samples" 2>&1 | tail -3
