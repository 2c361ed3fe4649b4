# single chunks near the end of the volume (shorter drain) against two chunks per increment throughout; same box
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 20 --warmup 3 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys,os; d=json.load(sys.stdin); print(os.environ.get('T2FIT_LIB','tapered')[-12:], d['config']['workload'][:28], 'kernel_ms', d['roofline']['kernel_ms'])"; }
{
for z in 32 64 128 256; do
run --shape $z 256 256
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_prev.so run --shape $z 256 256
done
} 2>&1 | tee gpurun_out/r02_exp60_tapered_take.txt
timeout -k 10 600 python tools/soak_kernel_variants.py 16 2>&1 | grep -v amdgpu.ids | tail -2
timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "large_volume or config or full_size or two_ranks" 2>&1 | tail -2
