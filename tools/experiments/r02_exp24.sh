cd $GRAFT_REPO_ROOT
for v in h2 i1 h2; do
  export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so
  python tools/kernel_ms.py $v
done | tee gpurun_out/r02_exp24.txt
