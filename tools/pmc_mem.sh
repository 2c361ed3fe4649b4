R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
SOLVER_ARGS="$1"; TAG="$2"
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT --output-format csv -d $R/gpurun_out/pmc_${TAG}_m -- python3 $R/bench.py $SOLVER_ARGS --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_${TAG}_m.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS --output-format csv -d $R/gpurun_out/pmc_${TAG}_n -- python3 $R/bench.py $SOLVER_ARGS --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_${TAG}_n.log 2>&1
