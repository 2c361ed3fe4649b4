# round 3, experiment 8: the global part of the correction pairs read past the vector cache (sc1 loads, L2-served) so that
# the L2 sees the lines in use: kernel time and rocprofv3 WRITE_SIZE / FETCH_SIZE against the plain-load build
cd $GRAFT_REPO_ROOT
{
python tools/kernel_ab.py sc1_loads --fit gaussian_rician --shape 256 256 256 --nte 8
python tools/kernel_ab.py sc1_loads --fit gaussian_rician --shape 256 256 256 --nte 8 --no_prior
python tools/kernel_ab.py sc1_loads --fit rician --shape 180 256 256 --nte 6
python tools/kernel_ab.py sc1_loads --fit gaussian_rician --shape 180 256 256 --nte 6
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_exp8_sc1_loads.txt
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_sc1_w -- python3 $R/bench.py --no-also --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_sc1_w.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_sc1_f -- python3 $R/bench.py --no-also --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_sc1_f.log 2>&1 &&
cd $R && python tools/pmc_summary.py sc1 persistent | tee -a gpurun_out/r03_exp8_sc1_loads.txt
