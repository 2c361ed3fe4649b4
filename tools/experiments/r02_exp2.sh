# round 2, experiment 2: echo-count specialised evaluation, sqrt without select, packed B -- bit-identity vs r01 + timing + stamps
set -o pipefail
cd $GRAFT_REPO_ROOT
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_r01.so python tools/map_digest.py > gpurun_out/r02_exp2_digest_r01.txt 2> gpurun_out/r02_exp2_digest_r01.err || { tail -5 gpurun_out/r02_exp2_digest_r01.err; exit 1; }
python tools/map_digest.py > gpurun_out/r02_exp2_digest_new.txt 2> gpurun_out/r02_exp2_digest_new.err || { tail -5 gpurun_out/r02_exp2_digest_new.err; exit 1; }
diff gpurun_out/r02_exp2_digest_r01.txt gpurun_out/r02_exp2_digest_new.txt && echo "DIGESTS IDENTICAL"
bash tools/sweep.sh T2FIT_NTE_SPECIAL "0 1" "--solver lbfgsb" 2>&1 | tee gpurun_out/r02_exp2_nte.txt
bash tools/sweep.sh T2FIT_REFILL_MIN "4 8 12 16" "--solver lbfgsb" 2>&1 | tee gpurun_out/r02_exp2_refill.txt
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_stamps.so python bench.py --no-also --cpu-seconds 0 --steps 2 --warmup 1 > gpurun_out/r02_exp2_stamps.json 2> gpurun_out/r02_exp2_stamps.err
grep "t2fit blocks" gpurun_out/r02_exp2_stamps.err | tail -11
