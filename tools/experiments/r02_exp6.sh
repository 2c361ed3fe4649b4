# round 2, experiment 6: host entry with masked-block skipping and a short first slab
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "context_api or host_entry or edge_inputs or torch_device_entry or volume_seam or cli" > gpurun_out/r02_exp6_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r02_exp6_pytest.log
[ $rc -ne 0 ] && exit $rc
for th in 1 2 4 8; do T2FIT_COPY_THREADS=$th python tools/host_entry_bench.py lbfgsb f64 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('threads $th', d['steady_state_ms'], d['seconds_fresh_outputs'][-1])"; done | tee gpurun_out/r02_exp6_threads.txt
for s in "lbfgsb f64" "lm f32" "lm f64" "loglin f64 gaussian"; do python tools/host_entry_bench.py $s 2>/dev/null; done | tee gpurun_out/r02_exp6_host_entry.jsonl
python bench.py --cpu-seconds 0 --no-also --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('kernel_ms', d['roofline']['kernel_ms'])"
