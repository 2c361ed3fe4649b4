cd $GRAFT_REPO_ROOT
for v in varVb default cur_fast r01; do
  unset T2FIT_LIB
  if [ $v != default ]; then export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so; fi
  python tools/kernel_ms.py $v
done | tee gpurun_out/r02_exp19.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stable_set or matches_reference or size_independent or host_entry or echo_train" 2>&1 | tail -3
