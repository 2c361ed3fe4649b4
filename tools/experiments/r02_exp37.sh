# counters of the one-wave-workgroup kernel held to 4 waves per CU against the 256-lane workgroup kernel (same occupancy)
R=$GRAFT_REPO_ROOT; cd $R
export T2FIT_WAVES_PER_CU=4
bash tools/pmc_passes.sh "--solver lbfgsb --no-also" w4 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_GDS --output-format csv -d $R/gpurun_out/pmc_w4_c -- python3 $R/bench.py --solver lbfgsb --no-also --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_w4_c.log 2>&1
unset T2FIT_WAVES_PER_CU
bash tools/pmc_passes.sh "--solver lbfgsb --no-also" w5 > /dev/null 2>&1
export T2FIT_WAVE_WG=0
bash tools/pmc_passes.sh "--solver lbfgsb --no-also" g256 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_GDS --output-format csv -d $R/gpurun_out/pmc_g256_c -- python3 $R/bench.py --solver lbfgsb --no-also --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_g256_c.log 2>&1
cd $R
python tools/pmc_summary.py w4 persistent > gpurun_out/r02_exp37_pmc_wave4.txt
python tools/pmc_summary.py w5 persistent > gpurun_out/r02_exp37_pmc_wave5.txt
python tools/pmc_summary.py g256 persistent > gpurun_out/r02_exp37_pmc_wg256.txt
paste gpurun_out/r02_exp37_pmc_wave4.txt gpurun_out/r02_exp37_pmc_wg256.txt | awk '{print $2, $4, $8}'
