# y stored scaled by 1/sqrt(y's) + reciprocal pivots in the subspace solve: against the previous library, same box
cd $GRAFT_REPO_ROOT
{
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_prev.so timeout -k 10 120 python tools/kernel_ms.py prev || exit 1
timeout -k 10 120 python tools/kernel_ms.py yhat_rcp || exit 1
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_prev.so timeout -k 10 120 python tools/kernel_ms.py prev || exit 1
timeout -k 10 120 python tools/kernel_ms.py yhat_rcp || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp43_yhat.txt
rm -f gpurun_out/parity_at_scale_suite.jsonl
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r02_exp43_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r02_exp43_pytest.log; exit $rc
