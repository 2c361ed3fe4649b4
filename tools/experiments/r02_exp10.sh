cd $GRAFT_REPO_ROOT
for v in default maxilp o2; do
  if [ $v = default ]; then unset T2FIT_LIB; else export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so; fi
  python bench.py --cpu-seconds 0 --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', 'lbfgsb', d['roofline']['kernel_ms'], 'lm32', d['also']['kernel_ms'], 'lm64', d['also_lm_f64']['kernel_ms'], 'loglin', d['also_loglin']['kernel_ms'])"
done | tee gpurun_out/r02_exp10_sched.txt
unset T2FIT_LIB
python tools/map_digest.py 32 128 128 > gpurun_out/r02_exp10_digest_default.txt 2>/dev/null
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_maxilp.so python tools/map_digest.py 32 128 128 > gpurun_out/r02_exp10_digest_maxilp.txt 2>/dev/null
diff gpurun_out/r02_exp10_digest_default.txt gpurun_out/r02_exp10_digest_maxilp.txt && echo "maxilp digests identical"
