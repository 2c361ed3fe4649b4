# s / s_0 with a fixed first component (no pivot bits, no selects, B s from 6 instead of 9 products): against the pivoted form
cd $GRAFT_REPO_ROOT
{
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_prev.so timeout -k 10 120 python tools/kernel_ms.py pivoted || exit 1
timeout -k 10 120 python tools/kernel_ms.py fixed_first || exit 1
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_prev.so timeout -k 10 120 python tools/kernel_ms.py pivoted || exit 1
timeout -k 10 120 python tools/kernel_ms.py fixed_first || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp51_fixed_first.txt
rm -f gpurun_out/parity_at_scale_suite.jsonl
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r02_exp51_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r02_exp51_pytest.log; exit $rc
