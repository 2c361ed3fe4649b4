# round 3, final state (after the lean log of the Rician lane): kernel-trace stats (headline alone, and the full default
# run), PMC passes of the Rician kernel, the other configurations, the default bench line
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03f -- python3 $R/bench.py --cpu-seconds 0 --no-also > $R/gpurun_out/prof_r03f.json 2> $R/gpurun_out/prof_r03f.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03f_all -- python3 $R/bench.py --cpu-seconds 0 > $R/gpurun_out/prof_r03f_all.json 2> $R/gpurun_out/prof_r03f_all.err &&
cd $R && bash tools/pmc_passes.sh "--fit rician --shape 180 256 256 --n-te 6 --no-also" ricianf > /dev/null 2>&1 &&
bash tools/pmc_extra.sh "--fit rician --shape 180 256 256 --n-te 6 --no-also" ricianf > /dev/null 2>&1 &&
python tools/pmc_summary.py ricianf persistent > gpurun_out/pmc_r03f_rician.txt &&
bash tools/bench_configs.sh > gpurun_out/bench_configs_r03f.jsonl 2>&1 &&
python bench.py > gpurun_out/bench_r03f.json 2> gpurun_out/bench_r03f.err; tail -c 600 gpurun_out/bench_r03f.json; grep -E "SQ_INSTS_VALU |WAVE_CYCLES|THREAD_CYCLES|ACTIVE_INST_VALU" gpurun_out/pmc_r03f_rician.txt
