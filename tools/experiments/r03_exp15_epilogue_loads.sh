# round 3: residuals_kernel with the parameter maps fetched beside the mask byte (two dependent round trips instead of three)
# A = tools/diag/libt2fit_base.so (before), B = the in-tree library; epilogue and step times of the headline and of LM float32
cd $GRAFT_REPO_ROOT
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "ms_per_step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms"], "epilogue", d.get("epilogue", d["roofline"].get("epilogue")))'
for rep in 1 2; do
T2FIT_LIB=$PWD/tools/diag/libt2fit_base.so python bench.py --no-also --cpu-seconds 0 2>/dev/null | python -c "$pick" A_lbfgsb &&
python bench.py --no-also --cpu-seconds 0 2>/dev/null | python -c "$pick" B_lbfgsb &&
T2FIT_LIB=$PWD/tools/diag/libt2fit_base.so python bench.py --no-also --cpu-seconds 0 --solver lm --precision f32 2>/dev/null | python -c "$pick" A_lm_f32 &&
python bench.py --no-also --cpu-seconds 0 --solver lm --precision f32 2>/dev/null | python -c "$pick" B_lm_f32 || exit 1
done
python tools/map_digest.py 64 256 256 > gpurun_out/r03_exp15_digest_new.txt 2>/dev/null
T2FIT_LIB=$PWD/tools/diag/libt2fit_base.so python tools/map_digest.py 64 256 256 > gpurun_out/r03_exp15_digest_old.txt 2>/dev/null; diff gpurun_out/r03_exp15_digest_old.txt gpurun_out/r03_exp15_digest_new.txt && echo "digests identical ($(wc -l < gpurun_out/r03_exp15_digest_new.txt) lines)"
