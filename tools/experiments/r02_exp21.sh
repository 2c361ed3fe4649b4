cd $GRAFT_REPO_ROOT
for v in e1 f4 f5 e1; do
  export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so
  python tools/kernel_ms.py $v
done | tee gpurun_out/r02_exp21.txt
