# kernel time against the size of a rank's share (strong scaling on 2 / 4 / 8 GPUs means 128 / 64 / 32 slices of the volume per rank)
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 10 --warmup 2 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['config']['workload'][:40], 'kernel_ms', d['roofline']['kernel_ms'], 'step_ms', d['ms_per_step'], 'Mvoxel/s', d['value'])"; }
{
run --shape 256 256 256
run --shape 128 256 256
run --shape 64 256 256
run --shape 32 256 256
run --shape 16 256 256
} 2>&1 | tee gpurun_out/r02_exp46_share_size.txt
