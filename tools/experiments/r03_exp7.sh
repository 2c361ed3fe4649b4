# round 3, experiment 7: i0e of a wave whose lanes need different series in ONE 30-step loop with a per-lane coefficient
# pick (instead of both series one after the other for every lane): Rician kernels against the round-2 library
cd $GRAFT_REPO_ROOT
{
for args in "--fit rician --shape 180 256 256 --nte 6" "--fit rician --shape 256 256 256 --nte 8" "--fit rician --shape 180 256 256 --nte 3" \
            "--fit rician --shape 180 256 256 --nte 6 --no_prior" "--fit rician --shape 180 256 256 --nte 6 --legacy" \
            "--fit rician --shape 8 256 256 --nte 9" "--fit rician --shape 8 256 256 --nte 17"; do
  T2FIT_LIB=tools/diag/libt2fit_r02.so python tools/kernel_ab.py r02 $args
  python tools/kernel_ab.py r03 $args
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_exp7_i0e_by_lane.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest3.log 2>&1; tail -4 gpurun_out/r03_gputest3.log
