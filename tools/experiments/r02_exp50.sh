# 2-parameter lane: three waves per SIMD (168 registers, 29-45 spilled to scratch, ten waves per CU) against two (eight)
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 10 --warmup 2 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys,os; d=json.load(sys.stdin); print(os.environ.get('T2FIT_LIB','product_hint3')[-20:], d['config']['workload'][:60], 'kernel_ms', d['roofline']['kernel_ms'])"; }
{
for rep in 1 2; do
run --shape 180 256 256 --n-te 6 --fit gaussian
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_prev.so run --shape 180 256 256 --n-te 6 --fit gaussian
done
run --shape 256 256 256 --n-te 8 --fit gaussian
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_prev.so run --shape 256 256 256 --n-te 8 --fit gaussian
} 2>&1 | tee gpurun_out/r02_exp50_2par_three_waves.txt
