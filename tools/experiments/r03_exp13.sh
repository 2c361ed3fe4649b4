# round 3, experiment 13 (not in the product): Cephes' Clenshaw recurrence of i0e with the product fused into the difference
# (fma(z, b1, -b2) + c: two instead of three operations per step) against the product library, which evaluates it as the
# reference's scipy does (a product, a difference, a sum): kernel time and agreement with the live oracle
cd $GRAFT_REPO_ROOT
{
python tools/kernel_ab.py product --fit rician --shape 180 256 256 --nte 6
T2FIT_LIB=tools/diag/libt2fit_i0efma.so python tools/kernel_ab.py i0e_fma --fit rician --shape 180 256 256 --nte 6
python tools/rician_log_parity.py product
T2FIT_LIB=tools/diag/libt2fit_i0efma.so python tools/rician_log_parity.py i0e_fma
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_exp13_i0e_fma.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest6.log 2>&1; tail -4 gpurun_out/r03_gputest6.log
