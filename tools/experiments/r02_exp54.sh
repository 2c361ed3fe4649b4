# LM float32 with staged results at three waves per SIMD (no register spills) against four, and against the unstaged library
R=$GRAFT_REPO_ROOT; cd $R
lm() { python bench.py --solver lm --precision $1 --steps 10 --warmup 2 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys,os; d=json.load(sys.stdin); print(os.environ.get('T2FIT_LIB','staged_4waves')[-16:], 'lm', sys.argv[1], 'kernel_ms', d['roofline']['kernel_ms'])" $1; }
{
for rep in 1 2; do
lm f32; T2FIT_LIB=$R/tools/diag/libt2fit_lm3.so lm f32; T2FIT_LIB=$R/tools/diag/libt2fit_prev.so lm f32
done
} 2>&1 | tee gpurun_out/r02_exp54_lm_f32_waves.txt
cd /tmp; export TMPDIR=/tmp
export T2FIT_LIB=$R/tools/diag/libt2fit_lm3.so
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_lm3_w -- python3 $R/bench.py --solver lm --precision f32 --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
(cd $R && python tools/pmc_summary.py lm3 persistent | awk '{print "lmf32 three waves", $2, $4}') | tee -a $R/gpurun_out/r02_exp54_lm_f32_waves.txt
