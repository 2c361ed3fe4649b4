# round 3, experiment 6: refill batch and chunk take re-tuned at eight waves per CU (they were tuned at four to six)
cd $GRAFT_REPO_ROOT
{
for r in 1 4 8 12 16 24; do
  T2FIT_REFILL_MIN=$r python tools/kernel_ab.py refill$r --fit gaussian_rician --shape 256 256 256 --nte 8
done
for t in 1 2 3 4; do
  T2FIT_TAKE=$t python tools/kernel_ab.py take$t --fit gaussian_rician --shape 256 256 256 --nte 8
done
for r in 4 8 16; do
  T2FIT_REFILL_MIN=$r python tools/kernel_ab.py refill$r --fit rician --shape 180 256 256 --nte 6
done
for r in 4 8 16; do
  T2FIT_REFILL_MIN=$r python tools/kernel_ab.py refill$r --fit gaussian --shape 180 256 256 --nte 6
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_exp6_refill_take.txt
