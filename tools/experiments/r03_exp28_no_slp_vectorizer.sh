# round 3: the library without the SLP vectorizer (-fno-slp-vectorize): the float32 LM evaluation is otherwise packed into
# v_pk_fma_f32 / v_pk_mul_f32 by the compiler at ten v_mov per echo to arrange the pairs.  P = in-tree, N = tools/diag/libt2fit_noslp.so
cd $GRAFT_REPO_ROOT
D=$PWD/tools/diag
run() { python tools/kernel_ab.py P "$@" 2>/dev/null | tail -1 && T2FIT_LIB=$D/libt2fit_noslp.so python tools/kernel_ab.py N "$@" 2>/dev/null | tail -1; }
run --solver lm --precision f32 && run --solver lm --precision f64 && run --solver lm --precision f32 --shape 180 256 256 --nte 6 && run --solver lm --precision f32 --fit gaussian --shape 180 256 256 --nte 6 &&
run --solver loglin --fit gaussian && run && run --fit gaussian --shape 180 256 256 --nte 6 && run --fit rician --shape 180 256 256 --nte 6 && run --solver lm --precision f32
