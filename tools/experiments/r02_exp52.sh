# min / max / abs of the L-BFGS-B lane as single instructions against compare + selects; same box
cd $GRAFT_REPO_ROOT
{
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_prev.so timeout -k 10 120 python tools/kernel_ms.py select_minmax || exit 1
timeout -k 10 120 python tools/kernel_ms.py native_minmax || exit 1
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_prev.so timeout -k 10 120 python tools/kernel_ms.py select_minmax || exit 1
timeout -k 10 120 python tools/kernel_ms.py native_minmax || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp52_native_minmax.txt
rm -f gpurun_out/parity_at_scale_suite.jsonl
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r02_exp52_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r02_exp52_pytest.log; exit $rc
