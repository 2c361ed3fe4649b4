# round 3: shared i0e loop of the Rician lane: every lane fetches its series' coefficients by vector loads from its own table
# pointer (one VMEM instruction per coefficient) instead of scalar loads of both tables + a per-lane pick (six vector
# instructions per coefficient on gfx950: one scalar operand per instruction).  P = tools/diag/libt2fit_p.so, B = in-tree.
cd $GRAFT_REPO_ROOT
D=$PWD/tools/diag
run() { T2FIT_LIB=$D/libt2fit_p.so python tools/kernel_ab.py P "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py B "$@" 2>/dev/null | tail -1; }
run --fit rician --shape 180 256 256 --nte 6 && run --fit rician && run --fit rician --legacy --shape 180 256 256 --nte 6 && run --fit rician --shape 64 256 256 --nte 3 &&
run --fit rician --shape 64 256 256 --nte 7 --extras && run --fit rician --shape 64 256 256 --nte 5 --no_prior && run --fit rician --shape 8 256 256 --nte 9 && run --fit rician --shape 180 256 256 --nte 6 &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rician or large_volume_kernels or stable_set or bad_samples" 2>&1 | tail -3
