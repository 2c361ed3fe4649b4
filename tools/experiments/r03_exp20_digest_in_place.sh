# round 3: (b2) build_b: ring slot stepped unconditionally, global part of the ring addressed as scalar base + 32-bit offset;
# (B) b2 + digest(): line-search state updated in place under the execution mask instead of on a copy kept by 13 selects.
# A = tools/diag/libt2fit_base.so (before both).  Digests must be equal.
cd $GRAFT_REPO_ROOT
A=$PWD/tools/diag/libt2fit_base.so; B2=$PWD/tools/diag/libt2fit_b2.so
run() { T2FIT_LIB=$A python tools/kernel_ab.py A "$@" 2>/dev/null | tail -1 && T2FIT_LIB=$B2 python tools/kernel_ab.py b2 "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py B "$@" 2>/dev/null | tail -1; }
run && run --no_prior && run --shape 180 256 256 --nte 6 && run --fit gaussian --shape 180 256 256 --nte 6 &&
run --fit rician --shape 180 256 256 --nte 6 && run --shape 64 256 256 --nte 7 --extras && run --shape 8 256 256 --nte 9 && run &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_difference or large_volume_kernels or stable_set or echo_train or bad_samples or edge_inputs or options" 2>&1 | tail -3
