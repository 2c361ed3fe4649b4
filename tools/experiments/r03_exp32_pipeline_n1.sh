# round 3: N = 1, consecutive steps rotating over 1 / 2 / 3 streams (the drain of a launch, 0.8 ms, overlapped by the next launch?)
cd $GRAFT_REPO_ROOT
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "ms_per_step", d["ms_per_step"], "median", d["ms_per_step_median"], "value", d["value"], "kernel", d["roofline"]["kernel_ms"])'
for rep in 1 2; do
python bench.py --no-also --cpu-seconds 0 --steps 40 2>/dev/null | python -c "$pick" one_stream &&
python bench.py --no-also --cpu-seconds 0 --steps 40 --pipeline on 2>/dev/null | python -c "$pick" two_streams &&
python bench.py --no-also --cpu-seconds 0 --steps 40 --pipeline on --pipeline-streams 3 2>/dev/null | python -c "$pick" three_streams &&
python bench.py --no-also --cpu-seconds 0 --steps 40 --pipeline on --pipeline-streams 4 2>/dev/null | python -c "$pick" four_streams || exit 1
done
