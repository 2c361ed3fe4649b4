# round 2, experiment 1 (GPU box, repo root): digest()/begin() split -- no-regression check, begin() parking sweep, block stamps
set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lbfgsb_matches_reference or size_independent or notebook" > gpurun_out/r02_exp1_pytest.log 2>&1 || { tail -20 gpurun_out/r02_exp1_pytest.log; exit 1; }
tail -2 gpurun_out/r02_exp1_pytest.log
bash tools/sweep.sh T2FIT_PARK_MIN "1 8 16 24 32 48" "--solver lbfgsb" > gpurun_out/r02_exp1_park_sweep.txt 2>&1
cat gpurun_out/r02_exp1_park_sweep.txt
T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_stamps.so python bench.py --no-also --cpu-seconds 0 --steps 2 --warmup 1 > gpurun_out/r02_exp1_stamps.json 2> gpurun_out/r02_exp1_stamps.err
grep "t2fit blocks" gpurun_out/r02_exp1_stamps.err | tail -11
T2FIT_PARK_MIN=32 T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_stamps.so python bench.py --no-also --cpu-seconds 0 --steps 2 --warmup 1 > gpurun_out/r02_exp1_stamps_park32.json 2> gpurun_out/r02_exp1_stamps_park32.err
grep "t2fit blocks" gpurun_out/r02_exp1_stamps_park32.err | tail -11
