# one-wave workgroups (five waves per CU) against 256-lane workgroups (four), same box, same library
cd $GRAFT_REPO_ROOT
{
for rep in 1 2; do
T2FIT_WAVE_WG=0 timeout -k 10 120 python tools/kernel_ms.py wg256 || exit 1
T2FIT_WAVE_WG=1 timeout -k 10 120 python tools/kernel_ms.py wg64x5 || exit 1
done
T2FIT_WAVE_WG=1 T2FIT_REFILL_MIN=4 timeout -k 10 120 python tools/kernel_ms.py wg64x5_refill4 || exit 1
T2FIT_WAVE_WG=1 T2FIT_REFILL_MIN=16 timeout -k 10 120 python tools/kernel_ms.py wg64x5_refill16 || exit 1
} 2>&1 | tee gpurun_out/r02_exp29_wave_wg.txt
T2FIT_WAVE_WG=0 timeout -k 10 300 python tools/map_digest.py > gpurun_out/r02_exp29_digest_wg256.txt 2>&1 || exit 1
T2FIT_WAVE_WG=1 timeout -k 10 300 python tools/map_digest.py > gpurun_out/r02_exp29_digest_wg64.txt 2>&1 || exit 1
diff gpurun_out/r02_exp29_digest_wg256.txt gpurun_out/r02_exp29_digest_wg64.txt && echo "map digests identical" | tee -a gpurun_out/r02_exp29_wave_wg.txt
