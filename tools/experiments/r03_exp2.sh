# round 3, experiment 2: where does the Rician lane's time go after the loop rewrite?  PMC passes on the cfg3-size
# volume (256x256x180 x 6 TE), and waves-per-CU sweeps (4 / 5 / 6) of the Rician and of the headline kernel: what
# would a seventh and eighth wave per CU be worth?
cd $GRAFT_REPO_ROOT
{
for w in 4 5 6; do
  T2FIT_WAVES_PER_CU=$w python tools/kernel_ab.py waves$w --fit rician --shape 180 256 256 --nte 6
  T2FIT_WAVES_PER_CU=$w python tools/kernel_ab.py waves$w --fit gaussian_rician --shape 256 256 256 --nte 8
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_exp2_waves.txt
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
A="--fit rician --shape 180 256 256 --n-te 6 --steps 2 --warmup 1 --cpu-seconds 0 --no-also"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_r03ric_a -- python3 $R/bench.py $A > $R/gpurun_out/pmc_r03ric_a.log 2>&1 &&
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d $R/gpurun_out/pmc_r03ric_b -- python3 $R/bench.py $A > $R/gpurun_out/pmc_r03ric_b.log 2>&1 &&
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc_r03ric_c -- python3 $R/bench.py $A > $R/gpurun_out/pmc_r03ric_c.log 2>&1 &&
cd $R && python tools/pmc_summary.py r03ric fit_ > gpurun_out/r03_pmc_rician.txt && cat gpurun_out/r03_pmc_rician.txt
