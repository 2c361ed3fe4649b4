# round 3: eval(): the forward-difference points as the plain x + h when every lane of the wave has them inside the box (scipy's
# step rules only for a wave with a lane at a bound); build_b(): first fetch unguarded, one register set less to clear.
# P = tools/diag/libt2fit_p.so (before), B = in-tree.  Digests must be equal.
cd $GRAFT_REPO_ROOT
D=$PWD/tools/diag
run() { T2FIT_LIB=$D/libt2fit_p.so python tools/kernel_ab.py P "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py B "$@" 2>/dev/null | tail -1; }
run && run --no_prior && run --shape 180 256 256 --nte 6 && run --fit gaussian --shape 180 256 256 --nte 6 && run --fit gaussian --no_prior && run --fit rician --shape 180 256 256 --nte 6 &&
run --shape 64 256 256 --nte 7 --extras && run --shape 8 256 256 --nte 9 && run &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_difference or large_volume_kernels or stable_set or echo_train or bad_samples or edge_inputs or options or matches_reference" 2>&1 | tail -3
