# usage: bash tools/sweep.sh ENVVAR "v1 v2 ..." "bench args"
for v in $2; do
  export $1=$v
  python bench.py $3 --steps 6 --warmup 2 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys,os; d=json.load(sys.stdin); print('$1=$v', d['config']['solver'], d['dtype'], 'kernel_ms', d['roofline']['kernel_ms'], 'step_ms', d['ms_per_step'])"
done
