cd $GRAFT_REPO_ROOT
export T2FIT_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1
for extra in "--scaling weak" "--scaling strong --no-gather" "--scaling strong --partition slab --pipeline off" "--scaling weak --reserve-cus 8"; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --shape 40 128 128 --no-also --cpu-seconds 0 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(sys.argv[1], '| value', d['value'], 'scaling', d['scaling'], 'pipelined', d['config']['steps_pipelined_over_two_streams'], 'reserve', d['config']['cus_left_free_for_rccl'], 'ab', d.get('reserve_cus_ab',{}).get('cus_left_free_for_rccl'), 'equal', d['gathered_maps_equal_single_gpu_fit'])" "$extra" || exit 1
done
