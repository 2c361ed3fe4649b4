# where the dispatcher puts the waves of the real kernel, per workgroup shape (diagnostic build)
cd $GRAFT_REPO_ROOT
export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_wgshape.so T2FIT_PLACEMENT=1
{
echo "== 64 lanes x 4 per CU";  T2FIT_WAVE_WG=1 T2FIT_WAVES_PER_CU=4 timeout -k 10 120 python tools/kernel_ms.py regs_wg64x4 2>&1 | sort | uniq -c | sort -rn | head -8
echo "== 128 lanes x 2 per CU"; T2FIT_WAVE_WG=3 timeout -k 10 120 python tools/kernel_ms.py regs_wg128x2 2>&1 | sort | uniq -c | sort -rn | head -8
echo "== 256 lanes x 1 per CU"; T2FIT_WAVE_WG=2 timeout -k 10 120 python tools/kernel_ms.py regs_wg256x1 2>&1 | sort | uniq -c | sort -rn | head -8
echo "== 64 lanes x 5 per CU";  T2FIT_WAVE_WG=1 timeout -k 10 120 python tools/kernel_ms.py regs_wg64x5 2>&1 | sort | uniq -c | sort -rn | head -8
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp40_placement.txt
