# round 3: the digest() in-place variant of exp21 (tools/experiments/patches/r03_exp21_digest_in_place.patch) once more, on top
# of the restructured main loop (exp22).  P = tools/diag/libt2fit_p.so (HEAD), B = in-tree with the patch.
cd $GRAFT_REPO_ROOT
D=$PWD/tools/diag
run() { T2FIT_LIB=$D/libt2fit_p.so python tools/kernel_ab.py P "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py B "$@" 2>/dev/null | tail -1; }
run && run --no_prior && run --shape 180 256 256 --nte 6 && run --fit gaussian --shape 180 256 256 --nte 6 && run --fit rician --shape 180 256 256 --nte 6 && run
