# round 3: is a v_mul_f64 / v_add_f64 really dearer than a v_fma_f64 on gfx950 (tools/diag/f64_issue_probe.hip: 1.6 x)?  In situ:
# the Rician lane's Clenshaw recurrence (a product, a difference, a sum: three roundings) with each of the three issued as a
# v_fma_f64 (a*b + 0, b*(-1) + a, a*1 + c) -- same three roundings, same bits.  P = in-tree, F = tools/diag/libt2fit_fmaform.so (-DT2_FMA_FORM)
cd $GRAFT_REPO_ROOT
D=$PWD/tools/diag
run() { python tools/kernel_ab.py P "$@" 2>/dev/null | tail -1 && T2FIT_LIB=$D/libt2fit_fmaform.so python tools/kernel_ab.py F "$@" 2>/dev/null | tail -1; }
run --fit rician --shape 180 256 256 --nte 6 && run --fit rician && run --fit rician --shape 64 256 256 --nte 3 && run --fit rician --shape 180 256 256 --nte 6
