# round 3, experiment 5: LM float32 with the Jacobian accumulation on packed math (two echoes per instruction) against the
# round-2 library, same box; T2 checksum-level agreement is checked by the LM tests of the suite (not bit-identical: the
# sums are formed in another order)
cd $GRAFT_REPO_ROOT
{
for args in "--solver lm --precision f32 --fit gaussian_rician --shape 256 256 256 --nte 8" "--solver lm --precision f32 --fit gaussian --shape 180 256 256 --nte 6" \
            "--solver lm --precision f32 --fit gaussian_rician --shape 180 256 256 --nte 6" "--solver lm --precision f32 --fit gaussian_rician --shape 256 256 256 --nte 7" \
            "--solver lm --precision f64 --fit gaussian_rician --shape 256 256 256 --nte 8"; do
  T2FIT_LIB=tools/diag/libt2fit_r02.so python tools/kernel_ab.py r02 $args
  python tools/kernel_ab.py r03 $args
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_exp5_lm_packed.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "lm or phantom or out_maps or shared_volume or loglin" 2>&1 | tail -5
