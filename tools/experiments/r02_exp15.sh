cd $GRAFT_REPO_ROOT
for v in r01_nte varVb nosplit_branchy_fast varVa r01_nte; do
  export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so
  python tools/kernel_ms.py $v
done | tee gpurun_out/r02_exp15_bisect.txt
