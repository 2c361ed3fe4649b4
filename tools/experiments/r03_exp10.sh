# round 3, experiment 10: cycle-stamped build (-DT2_PHASE_STAMPS: 256-lane workgroups, four waves per CU) -- where do a
# wave's cycles go in the round-3 solver code, for the headline objective and for the Rician likelihood?
cd $GRAFT_REPO_ROOT
{
T2FIT_LIB=tools/diag/libt2fit_stamps.so python tools/kernel_ab.py stamps --fit gaussian_rician --shape 256 256 256 --nte 8 --reps 1
T2FIT_LIB=tools/diag/libt2fit_stamps.so python tools/kernel_ab.py stamps --fit rician --shape 180 256 256 --nte 6 --reps 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_exp10_block_stamps.txt
