cd $GRAFT_REPO_ROOT
show() { python -c "import json,sys; d=json.load(sys.stdin); print(sys.argv[1], 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d['config']['steps_pipelined_over_two_streams'])" "$1"; }
{
python bench.py --cpu-seconds 0 --no-also 2>/dev/null | show "256^3 default"
python bench.py --cpu-seconds 0 --no-also --pipeline on 2>/dev/null | show "256^3 pipelined"
python bench.py --cpu-seconds 0 --no-also --shape 32 256 256 2>/dev/null | show "32 slices default"
python bench.py --cpu-seconds 0 --no-also --shape 32 256 256 --pipeline on 2>/dev/null | show "32 slices pipelined"
} 2>&1 | tee gpurun_out/r02_exp48_bench_pipeline.txt

