cd $GRAFT_REPO_ROOT
( cd tools/diag && for a in "1280 30720" "1280 32000" "1280 32768" "1024 30720" "1024 38400" "2048 20480" "1792 22528"; do timeout -k 5 60 ./wave_placement_probe $a || exit 1; done ) 2>&1 | tee gpurun_out/r02_exp34_wave_placement.txt
{
for cap in 4 5; do
T2FIT_WAVES_PER_CU=$cap timeout -k 10 120 python tools/kernel_ms.py wave_wg_cap$cap || exit 1
done
T2FIT_WAVE_WG=0 timeout -k 10 120 python tools/kernel_ms.py wg256 || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp34_wave_cap.txt
