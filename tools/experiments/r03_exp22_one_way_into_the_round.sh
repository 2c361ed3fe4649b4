# round 3: persistent kernel loop with ONE way into the evaluation round (the rare "all new voxels ended at once" case loops
# inside the refill block instead of `continue`-ing around the round; the T2FIT_PARK_MIN switch is gone): the solver state
# has one version per trip and the ~100 register copies per round between two register sets disappear (252 -> 219 VGPR).
# P = tools/diag/libt2fit_p.so (before), B = in-tree.  Digests must be equal.
cd $GRAFT_REPO_ROOT
D=$PWD/tools/diag
run() { T2FIT_LIB=$D/libt2fit_p.so python tools/kernel_ab.py P "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py B "$@" 2>/dev/null | tail -1; }
run && run --no_prior && run --shape 180 256 256 --nte 6 && run --fit gaussian --shape 180 256 256 --nte 6 && run --fit gaussian --no_prior &&
run --fit rician --shape 180 256 256 --nte 6 && run --fit rician && run --shape 64 256 256 --nte 7 --extras && run --shape 8 256 256 --nte 9 && run --shape 20 64 64 --nte 6 --fit gaussian --no_prior &&
run --solver lm --precision f32 && run --solver lm --precision f64 && run --solver lm --precision f32 --shape 180 256 256 --nte 6 && run &&
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
