cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in r01 default nte0; do
  unset T2FIT_LIB T2FIT_NTE_SPECIAL
  if [ $v = r01 ]; then export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_r01.so; fi
  if [ $v = nte0 ]; then export T2FIT_NTE_SPECIAL=0; fi
  python tools/kernel_ms.py $v
done; done | tee gpurun_out/r02_exp11_ab.txt
