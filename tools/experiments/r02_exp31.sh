# does a second wave on a SIMD pay?  configurations whose one-wave workgroups fit 5 (3 parameters, 3 TE) or 7 (2 parameters,
# 6 TE) to a CU, against 256-lane workgroups (4 waves per CU); same box, same library
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 5 --warmup 1 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys,os; d=json.load(sys.stdin); print(os.environ.get('T2FIT_WAVE_WG'), d['config']['workload'][:60], 'kernel_ms', d['roofline']['kernel_ms'])"; }
{
for w in 0 1 0 1; do
export T2FIT_WAVE_WG=$w
run --shape 256 256 256 --n-te 3 --fit gaussian_rician || exit 1
run --shape 180 256 256 --n-te 6 --fit gaussian || exit 1
run --shape 256 256 256 --n-te 3 --fit gaussian || exit 1
done
} 2>&1 | tee gpurun_out/r02_exp31_two_waves.txt
