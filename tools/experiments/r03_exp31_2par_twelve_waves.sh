# round 3: the 2-parameter lane with the split ring too (its one ratio per pair in global memory: 160 B of pairs per lane in LDS):
# twelve waves per CU, three on every SIMD, instead of ten.  P = tools/diag/libt2fit_p.so (ten waves), B = in-tree.
cd $GRAFT_REPO_ROOT
D=$PWD/tools/diag
run() { T2FIT_LIB=$D/libt2fit_p.so python tools/kernel_ab.py P "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py B "$@" 2>/dev/null | tail -1; }
run --fit gaussian --shape 180 256 256 --nte 6 && run --fit gaussian --no_prior && run --fit gaussian && run --fit gaussian --shape 64 256 256 --nte 3 --extras &&
run --fit gaussian --shape 64 256 256 --nte 7 && run --fit gaussian --shape 180 256 256 --nte 6 --no_prior && run --fit gaussian --shape 180 256 256 --nte 6 &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "large_volume_kernels or stable_set or echo_train or config2" 2>&1 | tail -3
