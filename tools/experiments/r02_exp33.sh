# one-wave workgroups: full GPU suite with the new default, then the same kernel capped at 4 and 5 waves per CU
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity_at_scale_suite.jsonl
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r02_exp33_pytest.log 2>&1; rc=$?; tail -4 gpurun_out/r02_exp33_pytest.log; [ $rc -eq 0 ] || exit $rc
{
for cap in 4 5 3; do
T2FIT_WAVES_PER_CU=$cap timeout -k 10 120 python tools/kernel_ms.py wave_wg_cap$cap || exit 1
done
T2FIT_WAVE_WG=0 timeout -k 10 120 python tools/kernel_ms.py wg256 || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp33_wave_cap.txt
