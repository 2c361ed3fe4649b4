R=$GRAFT_REPO_ROOT; cd $R
export T2FIT_WAVES_PER_CU=4
bash tools/pmc_extra.sh "--solver lbfgsb --no-also" x4 > /dev/null 2>&1
unset T2FIT_WAVES_PER_CU
export T2FIT_WAVE_WG=0
bash tools/pmc_extra.sh "--solver lbfgsb --no-also" x256 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py x4 persistent > gpurun_out/r02_exp38_pmc_wave4.txt
python tools/pmc_summary.py x256 persistent > gpurun_out/r02_exp38_pmc_wg256.txt
paste gpurun_out/r02_exp38_pmc_wave4.txt gpurun_out/r02_exp38_pmc_wg256.txt | awk '{print $2, $4, $8}'
