# s stored as a direction (5 doubles per pair, 400 B per lane): six waves per CU; against the previous library, same box
cd $GRAFT_REPO_ROOT
{
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_prev.so timeout -k 10 120 python tools/kernel_ms.py prev_5waves || exit 1
timeout -k 10 120 python tools/kernel_ms.py direction_6waves || exit 1
T2FIT_WAVES_PER_CU=5 timeout -k 10 120 python tools/kernel_ms.py direction_capped_5waves || exit 1
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_prev.so timeout -k 10 120 python tools/kernel_ms.py prev_5waves || exit 1
timeout -k 10 120 python tools/kernel_ms.py direction_6waves || exit 1
T2FIT_WAVE_WG=0 timeout -k 10 120 python tools/kernel_ms.py direction_wg256 || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp44_direction.txt
rm -f gpurun_out/parity_at_scale_suite.jsonl
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r02_exp44_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r02_exp44_pytest.log; exit $rc
