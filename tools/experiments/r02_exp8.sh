cd $GRAFT_REPO_ROOT
python tools/overlap_check_rccl.py 2>/dev/null | grep reserve_cus | tee gpurun_out/r02_exp8_overlap_rccl.jsonl
python tools/overlap_check.py 2>/dev/null | grep reserve_cus | tee gpurun_out/r02_exp8_overlap_copy.jsonl
