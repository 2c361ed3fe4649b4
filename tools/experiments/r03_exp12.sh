# round 3, experiment 12 (not in the product): the Rician likelihood with an fdlibm-style log (< 1 ulp, 35 instructions)
# for log(i0e) instead of the device library's (double-double, 75 instructions): kernel time and, on 20 000 voxels of the bench
# distribution, agreement with the live oracle -- what the leaner function buys and what it costs
cd $GRAFT_REPO_ROOT
{
python tools/kernel_ab.py product --fit rician --shape 180 256 256 --nte 6
T2FIT_LIB=tools/diag/libt2fit_leanlog.so python tools/kernel_ab.py lean_log --fit rician --shape 180 256 256 --nte 6
python tools/kernel_ab.py product --fit rician --shape 256 256 256 --nte 8
T2FIT_LIB=tools/diag/libt2fit_leanlog.so python tools/kernel_ab.py lean_log --fit rician --shape 256 256 256 --nte 8
python tools/rician_log_parity.py product
T2FIT_LIB=tools/diag/libt2fit_leanlog.so python tools/rician_log_parity.py lean_log
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_exp12_lean_log.txt
