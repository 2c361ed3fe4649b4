# round 3: residuals_kernel as a loop over tiles with a bounded grid (T2FIT_RESIDUAL_WG_PER_CU workgroups per CU; 0 = one
# workgroup per tile, the old shape) against tools/diag/libt2fit_base.so; LM float32 steps (the epilogue is 15 % of them)
cd $GRAFT_REPO_ROOT
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "ms_per_step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms"], "epilogue", (d.get("epilogue") or d["roofline"]["epilogue"])["kernel_ms"])'
T2FIT_LIB=$PWD/tools/diag/libt2fit_base.so python bench.py --no-also --cpu-seconds 0 --solver lm --precision f32 2>/dev/null | python -c "$pick" base || exit 1
for w in 0 2 4 8 16 32 64; do
T2FIT_RESIDUAL_WG_PER_CU=$w python bench.py --no-also --cpu-seconds 0 --solver lm --precision f32 2>/dev/null | python -c "$pick" wg_per_cu_$w || exit 1
done
python bench.py --no-also --cpu-seconds 0 2>/dev/null | python -c "$pick" lbfgsb_default
python tools/map_digest.py 64 256 256 > gpurun_out/r03_exp16_digest_new.txt 2>/dev/null
T2FIT_LIB=$PWD/tools/diag/libt2fit_base.so python tools/map_digest.py 64 256 256 > gpurun_out/r03_exp16_digest_old.txt 2>/dev/null; diff gpurun_out/r03_exp16_digest_old.txt gpurun_out/r03_exp16_digest_new.txt && echo "digests identical ($(wc -l < gpurun_out/r03_exp16_digest_new.txt) lines)"
