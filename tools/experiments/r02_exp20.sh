cd $GRAFT_REPO_ROOT
for v in default e1 e3 default; do
  unset T2FIT_LIB
  if [ $v != default ]; then export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so; fi
  python tools/kernel_ms.py $v
done | tee gpurun_out/r02_exp20.txt
