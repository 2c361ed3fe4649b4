# round 3, experiment 3: eight one-wave workgroups per CU for the three-parameter lanes (one number of each correction
# pair in global memory, 320 instead of 400 bytes of LDS per lane) against six (round-2 library and T2FIT_WAVES_PER_CU=6)
cd $GRAFT_REPO_ROOT
{
for args in "--fit gaussian_rician --shape 256 256 256 --nte 8" "--fit gaussian_rician --shape 256 256 256 --nte 8 --no_prior" \
            "--fit gaussian_rician --shape 180 256 256 --nte 6" "--fit rician --shape 180 256 256 --nte 6" \
            "--fit gaussian_rician --shape 256 256 256 --nte 8 --extras" "--fit gaussian_rician --shape 32 256 256 --nte 8" \
            "--fit gaussian --shape 180 256 256 --nte 6"; do
  T2FIT_LIB=tools/diag/libt2fit_r02.so python tools/kernel_ab.py r02 $args
  python tools/kernel_ab.py r03 $args
  T2FIT_WAVES_PER_CU=7 python tools/kernel_ab.py r03_7waves $args
  T2FIT_WAVES_PER_CU=6 python tools/kernel_ab.py r03_6waves $args
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_exp3_eight_waves.txt
