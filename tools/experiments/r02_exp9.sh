cd $GRAFT_REPO_ROOT
python tools/overlap_check2.py 2>/dev/null | grep reserve_cus | tee gpurun_out/r02_exp9_overlap2.jsonl
