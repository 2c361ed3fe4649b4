# the register-queue code in workgroups of 64 (capped at 4 per CU), 128 (2 per CU) and 256 lanes (1 per CU): same code,
# same occupancy, only the workgroup shape differs; and the LDS-queue 256-lane kernel
cd $GRAFT_REPO_ROOT
export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_wgshape.so
{
T2FIT_WAVE_WG=1 T2FIT_WAVES_PER_CU=4 timeout -k 10 120 python tools/kernel_ms.py regs_wg64x4 || exit 1
T2FIT_WAVE_WG=3 timeout -k 10 120 python tools/kernel_ms.py regs_wg128x2 || exit 1
T2FIT_WAVE_WG=2 timeout -k 10 120 python tools/kernel_ms.py regs_wg256x1 || exit 1
T2FIT_WAVE_WG=0 timeout -k 10 120 python tools/kernel_ms.py ldsq_wg256x1 || exit 1
T2FIT_WAVE_WG=1 timeout -k 10 120 python tools/kernel_ms.py regs_wg64x5 || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp39_wg_shape.txt
