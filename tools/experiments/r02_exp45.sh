cd $GRAFT_REPO_ROOT
{
for r in 4 6 8 12; do T2FIT_REFILL_MIN=$r timeout -k 10 120 python tools/kernel_ms.py refill_min=$r || exit 1; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp45_refill_6waves.txt
bash tools/profile_r02.sh r02e
