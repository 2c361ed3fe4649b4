# LM lanes: chunk results staged in LDS tiles and written as whole lines: time and HBM writes against the previous library
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "lm_ or lm or stream or context or host_entry" > gpurun_out/r02_exp53_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r02_exp53_pytest.log; [ $rc -eq 0 ] || exit $rc
lm() { python bench.py --solver lm --precision $1 --steps 10 --warmup 2 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys,os; d=json.load(sys.stdin); print(os.environ.get('T2FIT_LIB','staged')[-12:], 'lm', sys.argv[1], 'kernel_ms', d['roofline']['kernel_ms'])" $1; }
{
for rep in 1 2; do
lm f32; T2FIT_LIB=$R/tools/diag/libt2fit_prev.so lm f32
lm f64; T2FIT_LIB=$R/tools/diag/libt2fit_prev.so lm f64
done
} 2>&1 | tee gpurun_out/r02_exp53_lm_staging.txt
cd /tmp; export TMPDIR=/tmp
for spec in "lmf32:--solver lm --precision f32" "lmf64:--solver lm --precision f64" "lbfgsb:--solver lbfgsb --no-also"; do
  t=${spec%%:*}; args=${spec#*:}
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_st${t}_w -- python3 $R/bench.py $args --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_st${t}_f -- python3 $R/bench.py $args --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
  (cd $R && python tools/pmc_summary.py st${t} persistent | awk '{print "'$t'", $2, $4}') | tee -a $R/gpurun_out/r02_exp53_lm_staging.txt
done
