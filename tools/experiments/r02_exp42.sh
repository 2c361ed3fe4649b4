# non-temporal sample loads: kernel time and HBM write volume of the LM float32 and the L-BFGS-B kernels
R=$GRAFT_REPO_ROOT; cd $R
{
timeout -k 10 120 python tools/kernel_ms.py nt_loads_lbfgsb || exit 1
python bench.py --solver lm --precision f32 --steps 10 --warmup 2 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('lm f32 kernel_ms', d['roofline']['kernel_ms'])"
python bench.py --solver lm --precision f64 --steps 10 --warmup 2 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('lm f64 kernel_ms', d['roofline']['kernel_ms'])"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_exp42_nt_loads.txt
cd /tmp; export TMPDIR=/tmp
for spec in "lmf32:--solver lm --precision f32" "lmf64:--solver lm --precision f64" "lbfgsb:--solver lbfgsb --no-also"; do
  t=${spec%%:*}; args=${spec#*:}
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_nt${t}_w -- python3 $R/bench.py $args --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_nt${t}_f -- python3 $R/bench.py $args --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
  (cd $R && python tools/pmc_summary.py nt${t} persistent | awk '{print "'$t'", $2, $4}') | tee -a $R/gpurun_out/r02_exp42_nt_loads.txt
done
