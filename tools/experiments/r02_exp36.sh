cd $GRAFT_REPO_ROOT
{
for cap in 4 5; do
T2FIT_WAVES_PER_CU=$cap timeout -k 10 120 python tools/kernel_ms.py take4_wave_wg_cap$cap || exit 1
done
T2FIT_WAVE_WG=0 timeout -k 10 120 python tools/kernel_ms.py wg256 || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp36_take4.txt
T2FIT_WAVE_WG=0 timeout -k 10 300 python tools/map_digest.py > gpurun_out/r02_exp36_digest_wg256.txt 2>&1 || exit 1
T2FIT_WAVE_WG=1 timeout -k 10 300 python tools/map_digest.py > gpurun_out/r02_exp36_digest_wg64.txt 2>&1 || exit 1
diff gpurun_out/r02_exp36_digest_wg256.txt gpurun_out/r02_exp36_digest_wg64.txt && echo "map digests identical" | tee -a gpurun_out/r02_exp36_take4.txt
