cd $GRAFT_REPO_ROOT
for np in "" "--no-prior"; do
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_stamps.so python bench.py --no-also --cpu-seconds 0 --steps 2 --warmup 1 $np > gpurun_out/r02_exp7_stamps.json 2> gpurun_out/r02_exp7_stamps.err
echo "== $np"; grep "t2fit blocks" gpurun_out/r02_exp7_stamps.err | tail -11
done
python tools/overlap_check_rccl.py 2>&1 | tail -4 | tee gpurun_out/r02_exp7_overlap_rccl.jsonl
