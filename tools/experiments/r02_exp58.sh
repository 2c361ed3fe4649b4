# mid-size volumes: the generic small-chunk kernel (default below 2^20 voxels) against the one-wave-workgroup kernels
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 10 --warmup 2 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys,os; d=json.load(sys.stdin); print('small_volume<=', os.environ.get('T2FIT_SMALL_VOLUME','default'), d['config']['workload'][:28], 'kernel_ms', d['roofline']['kernel_ms'])"; }
{
for z in 2 4 8 12 16; do
run --shape $z 256 256
T2FIT_SMALL_VOLUME=65536 run --shape $z 256 256
done
run --shape 20 64 64 --n-te 6 --fit gaussian --no-prior
T2FIT_SMALL_VOLUME=0 run --shape 20 64 64 --n-te 6 --fit gaussian --no-prior
} 2>&1 | tee gpurun_out/r02_exp58_mid_size.txt
