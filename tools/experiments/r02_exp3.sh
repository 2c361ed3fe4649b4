# round 2, experiment 3: straight-line digest (three states side by side), -ffp-contract=off with explicit fma, vs r01
set -o pipefail
cd $GRAFT_REPO_ROOT
python tools/map_digest.py > gpurun_out/r02_exp3_digest_new.txt 2> gpurun_out/r02_exp3_digest_new.err || { tail -5 gpurun_out/r02_exp3_digest_new.err; exit 1; }
diff profiles/r02_map_digest_r01_library.txt gpurun_out/r02_exp3_digest_new.txt > gpurun_out/r02_exp3_digest_diff.txt; cat gpurun_out/r02_exp3_digest_diff.txt | head -40
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lbfgsb or notebook or echo_train or options_against or traces" > gpurun_out/r02_exp3_pytest.log 2>&1 || { tail -30 gpurun_out/r02_exp3_pytest.log; }
tail -3 gpurun_out/r02_exp3_pytest.log
bash tools/sweep.sh T2FIT_NTE_SPECIAL "0 1" "--solver lbfgsb" 2>&1 | tee gpurun_out/r02_exp3_nte.txt
python bench.py --cpu-seconds 0 --steps 6 --warmup 2 > gpurun_out/r02_exp3_bench.json 2>/dev/null; cat gpurun_out/r02_exp3_bench.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['roofline']['kernel_ms'], d['also']['kernel_ms'], d['also_loglin']['kernel_ms'])"
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_stamps.so python bench.py --no-also --cpu-seconds 0 --steps 2 --warmup 1 > gpurun_out/r02_exp3_stamps.json 2> gpurun_out/r02_exp3_stamps.err
grep "t2fit blocks" gpurun_out/r02_exp3_stamps.err | tail -11
python tools/parity_at_scale.py 20000 --all > gpurun_out/r02_exp3_parity_20k_all.json 2> gpurun_out/r02_exp3_parity.err || tail -5 gpurun_out/r02_exp3_parity.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_exp3_parity_20k_all.json'))
for k,v in d['configs'].items():
    if 'hip_lbfgsb_vs_reference' in v:
        a,b=v['hip_lbfgsb_vs_reference'],v['reference_vs_itself_one_ulp_exp']
        print(k, 'HIP', round(a['within_1ms'],4), 'floor', round(b['within_1ms'],4), 'p50', a['median_ms'], b['median_ms'], 'p99', a['p99_ms'], b['p99_ms'], 'ok', a['success_equal'], 'nit', a['nit_equal'])
PY
