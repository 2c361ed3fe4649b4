# round 3, experiment 11: config 5 streaming with mask-aware transfers (only the stretches of a volume that hold masked
# voxels cross PCIe, one 2-D copy per stretch): per-subject time with pinned I/O, against the plain full-volume copies
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "streamed or config5" 2>&1 | tail -4
{
python tools/stream_bench.py 16 lbfgsb f64
python tools/stream_bench.py 16 lm f32
python tools/stream_bench.py 16 loglin f64 gaussian
} 2>/dev/null | tee gpurun_out/r03_exp11_stream_mask_aware.jsonl
