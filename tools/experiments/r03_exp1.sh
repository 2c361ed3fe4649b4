# round 3, experiment 1: the Rician-likelihood lane as loops in one-wave workgroups (current library) against the
# round-2 library (unrolled evaluation, 256-lane workgroups), same box: kernel time and map digests; then the GPU suite
cd $GRAFT_REPO_ROOT
{
for args in "--fit rician --shape 180 256 256 --nte 6" "--fit rician --shape 256 256 256 --nte 8" "--fit rician --shape 180 256 256 --nte 3" \
            "--fit rician --shape 64 256 256 --nte 5" "--fit rician --shape 180 256 256 --nte 6 --no_prior" \
            "--fit gaussian_rician --shape 256 256 256 --nte 8" "--fit gaussian --shape 180 256 256 --nte 6"; do
  T2FIT_LIB=tools/diag/libt2fit_r02.so python tools/kernel_ab.py r02 $args
  python tools/kernel_ab.py r03 $args
done
python tools/kernel_ab.py r03 --fit rician --shape 180 256 256 --nte 6 --legacy
python tools/kernel_ab.py r03 --fit rician --shape 8 256 256 --nte 6
T2FIT_LIB=tools/diag/libt2fit_r02.so python tools/kernel_ab.py r02 --fit rician --shape 8 256 256 --nte 6
python tools/kernel_ab.py r03 --fit rician --shape 8 256 256 --nte 9
T2FIT_LIB=tools/diag/libt2fit_r02.so python tools/kernel_ab.py r02 --fit rician --shape 8 256 256 --nte 9
python tools/kernel_ab.py r03 --fit rician --shape 8 256 256 --nte 17
T2FIT_LIB=tools/diag/libt2fit_r02.so python tools/kernel_ab.py r02 --fit rician --shape 8 256 256 --nte 17
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_exp1_rician_loops.txt
