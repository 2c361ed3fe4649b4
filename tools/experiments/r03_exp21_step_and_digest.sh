# round 3: on top of the two-register-set build_b (P = tools/diag/libt2fit_p.so): (Q) the ring slot stepped past the last pair
# unconditionally; (R, in-tree) Q + digest(): line-search state updated in place under the execution mask instead of on a copy
# kept by 13 selects.  A = tools/diag/libt2fit_base.so (before the register sets).  Digests must be equal.
# (exp20 had also addressed the global part of the ring as scalar base + 32-bit offset: the base pointer then lives in spilled
#  scalar registers, two v_readlane + s_nop 4 per pair, and the headline went 11.27 -> 11.65 ms: dropped.)
cd $GRAFT_REPO_ROOT
D=$PWD/tools/diag
run() { T2FIT_LIB=$D/libt2fit_base.so python tools/kernel_ab.py A "$@" 2>/dev/null | tail -1 && T2FIT_LIB=$D/libt2fit_p.so python tools/kernel_ab.py P "$@" 2>/dev/null | tail -1 &&
        T2FIT_LIB=$D/libt2fit_q.so python tools/kernel_ab.py Q "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py R "$@" 2>/dev/null | tail -1; }
run && run --no_prior && run --shape 180 256 256 --nte 6 && run --fit gaussian --shape 180 256 256 --nte 6 &&
run --fit rician --shape 180 256 256 --nte 6 && run --shape 64 256 256 --nte 7 --extras && run --shape 8 256 256 --nte 9 && run &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_difference or large_volume_kernels or stable_set or echo_train or bad_samples or edge_inputs or options" 2>&1 | tail -3
