# usage (GPU box, repo root): bash tools/pmc_extra.sh "<bench args>" <tag>  -- instruction-cache and instruction-mix counters, own passes
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
SOLVER_ARGS="$1"; TAG="$2"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_${TAG}_c -- python3 $R/bench.py $SOLVER_ARGS --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_${TAG}_c.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 --output-format csv -d $R/gpurun_out/pmc_${TAG}_d -- python3 $R/bench.py $SOLVER_ARGS --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_${TAG}_d.log 2>&1
