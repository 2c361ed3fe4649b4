R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
SOLVER_ARGS="$1"; TAG="$2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -- python3 $R/bench.py $SOLVER_ARGS --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -- python3 $R/bench.py $SOLVER_ARGS --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_${TAG}_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_f -- python3 $R/bench.py $SOLVER_ARGS --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_${TAG}_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_w -- python3 $R/bench.py $SOLVER_ARGS --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/pmc_${TAG}_w.log 2>&1
ls $R/gpurun_out/pmc_${TAG}_*/*/ | head -20
