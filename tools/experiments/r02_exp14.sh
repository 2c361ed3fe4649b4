cd $GRAFT_REPO_ROOT
for v in r01_nte nosplit_branchy_fast varR1 varR2 varR3 r01_nte; do
  export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so
  python tools/kernel_ms.py $v
done | tee gpurun_out/r02_exp14_bisect.txt
