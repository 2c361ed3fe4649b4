# usage (GPU box): bash tools/sweep_refill.sh  -- kernel ms of the persistent fits against the refill batch and grid size
for s in "lm f32" "lbfgsb f64"; do set -- $s
  for r in 4 8 12 16 24 32; do
    T2FIT_REFILL_MIN=$r python bench.py --solver $1 --precision $2 --steps 5 --warmup 2 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 refill_min=$r kernel_ms', d['roofline']['kernel_ms'])"
  done
  for b in 1024 2048 4096 8192; do
    T2FIT_PERSISTENT_BLOCKS=$b python bench.py --solver $1 --precision $2 --steps 5 --warmup 2 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 blocks=$b kernel_ms', d['roofline']['kernel_ms'])"
  done
done
