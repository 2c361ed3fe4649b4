cd $GRAFT_REPO_ROOT
for v in lmw2 lmw3; do
  T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so python bench.py --cpu-seconds 0 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v nte', 'lm32', d['also']['kernel_ms'], 'lm64', d['also_lm_f64']['kernel_ms'])"
  T2FIT_NTE_SPECIAL=0 T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so python bench.py --cpu-seconds 0 --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v generic', 'lm32', d['also']['kernel_ms'], 'lm64', d['also_lm_f64']['kernel_ms'])"
done | tee gpurun_out/r02_exp28_lm.txt
