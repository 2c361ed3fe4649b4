# one-wave workgroups, samples and queue in registers, chunk read-ahead: against 256-lane workgroups; same box
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 5 --warmup 1 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys,os; d=json.load(sys.stdin); print(os.environ.get('T2FIT_WAVE_WG'), d['config']['workload'][:60], 'kernel_ms', d['roofline']['kernel_ms'])"; }
{
for w in 0 1 0 1; do
export T2FIT_WAVE_WG=$w
timeout -k 10 120 python tools/kernel_ms.py wave_wg=$w || exit 1
run --shape 180 256 256 --n-te 6 --fit gaussian || exit 1
run --shape 180 256 256 --n-te 6 --fit gaussian_rician || exit 1
done
T2FIT_WAVE_WG=1 T2FIT_REFILL_MIN=4 timeout -k 10 120 python tools/kernel_ms.py wave_wg=1_refill4 || exit 1
T2FIT_WAVE_WG=1 T2FIT_REFILL_MIN=12 timeout -k 10 120 python tools/kernel_ms.py wave_wg=1_refill12 || exit 1
T2FIT_WAVE_WG=1 T2FIT_REFILL_MIN=16 timeout -k 10 120 python tools/kernel_ms.py wave_wg=1_refill16 || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp32_wave_wg.txt
T2FIT_WAVE_WG=0 timeout -k 10 300 python tools/map_digest.py > gpurun_out/r02_exp32_digest_wg256.txt 2>&1 || exit 1
T2FIT_WAVE_WG=1 timeout -k 10 300 python tools/map_digest.py > gpurun_out/r02_exp32_digest_wg64.txt 2>&1 || exit 1
diff gpurun_out/r02_exp32_digest_wg256.txt gpurun_out/r02_exp32_digest_wg64.txt && echo "map digests identical" | tee -a gpurun_out/r02_exp32_wave_wg.txt
