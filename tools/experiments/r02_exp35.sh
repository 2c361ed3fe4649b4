cd $GRAFT_REPO_ROOT
{
for cap in 4 5; do
T2FIT_WAVES_PER_CU=$cap timeout -k 10 120 python tools/kernel_ms.py hint1_wave_wg_cap$cap || exit 1
done
T2FIT_WAVE_WG=0 timeout -k 10 120 python tools/kernel_ms.py wg256 || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp35_wave_hint1.txt
