cd $GRAFT_REPO_ROOT
for v in r01 r01_nte cur_fast nosplit_fast nosplit_branchy_fast branchy_fast default r01; do
  unset T2FIT_LIB
  if [ $v != default ]; then export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so; fi
  python tools/kernel_ms.py $v
done | tee gpurun_out/r02_exp13_bisect.txt
