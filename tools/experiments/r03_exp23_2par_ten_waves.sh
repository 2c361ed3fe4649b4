# round 3: the 2-parameter lane at three waves per SIMD (registers: 155-168 since the loop restructuring of exp22; LDS: 240 B of
# pairs per lane = ten one-wave workgroups per CU).  P = tools/diag/libt2fit_p.so (eight waves per CU), B = in-tree (-DT2_WAVE_HINT_2PAR=3).
cd $GRAFT_REPO_ROOT
D=$PWD/tools/diag
run() { T2FIT_LIB=$D/libt2fit_p.so python tools/kernel_ab.py P "$@" 2>/dev/null | tail -1 && python tools/kernel_ab.py B "$@" 2>/dev/null | tail -1; }
run --fit gaussian --shape 180 256 256 --nte 6 && run --fit gaussian --no_prior && run --fit gaussian && run --fit gaussian --shape 64 256 256 --nte 3 --extras &&
run --fit gaussian --shape 64 256 256 --nte 7 && run --fit gaussian --shape 180 256 256 --nte 6 --no_prior && run --fit gaussian --shape 180 256 256 --nte 6
