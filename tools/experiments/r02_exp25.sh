cd $GRAFT_REPO_ROOT
for v in default j1 default; do
  unset T2FIT_LIB
  if [ $v != default ]; then export T2FIT_LIB=$GRAFT_REPO_ROOT/tools/diag/libt2fit_$v.so; fi
  python tools/kernel_ms.py $v
done | tee gpurun_out/r02_exp25.txt
unset T2FIT_LIB
bash tools/pmc_passes.sh "--solver lbfgsb --no-also" lbx > /dev/null 2>&1
python tools/pmc_summary.py lbx persistent | awk '{print $2,$3,$4}' | tee gpurun_out/r02_exp25_pmc.txt
