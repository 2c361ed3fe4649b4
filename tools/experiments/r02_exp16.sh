cd $GRAFT_REPO_ROOT
for v in varVb varVd varVe nosplit_branchy_fast varVb; do
  export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so
  python tools/kernel_ms.py $v
done | tee gpurun_out/r02_exp16_bisect.txt
