# round 3: the three displaced square roots of an echo (3-parameter least-squares evaluation) started from the base point's
# reciprocal root (t2_sqrt_near: 6 instead of 11 instructions, same bits).  A = tools/diag/libt2fit_base.so (before),
# B = in-tree.  Kernel time + SHA-256 of the four maps per configuration (tools/kernel_ab.py): the digests must be equal.
cd $GRAFT_REPO_ROOT
A=$PWD/tools/diag/libt2fit_base.so
run() { T2FIT_LIB=$A python tools/kernel_ab.py A "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py B "$@" 2>/dev/null | tail -1; }
run && run --no_prior && run --shape 180 256 256 --nte 6 && run --shape 180 256 256 --nte 3 && run --shape 64 256 256 --nte 7 --extras &&
run --shape 64 256 256 --nte 5 --no_prior && run --shape 8 256 256 --nte 8 && run --shape 8 256 256 --nte 9 && run &&
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_difference or large_volume_kernels_equal or stable_set or echo_train" 2>&1 | tail -3
