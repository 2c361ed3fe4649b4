# round 3, experiment 9: (a) ring slot stepped along in build_b instead of (head + p) % 10 per pair (current library against
# the snapshot before it, tools/diag/libt2fit_r03a.so); (b) coefficients per trip of the shared i0e loop: 3 / 5 / 6 / 10
cd $GRAFT_REPO_ROOT
{
for args in "--fit gaussian_rician --shape 256 256 256 --nte 8" "--fit gaussian --shape 180 256 256 --nte 6" "--fit rician --shape 180 256 256 --nte 6"; do
  T2FIT_LIB=tools/diag/libt2fit_r03a.so python tools/kernel_ab.py before $args
  python tools/kernel_ab.py slot_stepped_chunk5 $args
done
for k in 3 6 10; do
  T2FIT_LIB=tools/diag/libt2fit_i0e$k.so python tools/kernel_ab.py chunk$k --fit rician --shape 180 256 256 --nte 6
  T2FIT_LIB=tools/diag/libt2fit_i0e$k.so python tools/kernel_ab.py chunk$k --fit rician --shape 256 256 256 --nte 8
done
python tools/kernel_ab.py chunk5 --fit rician --shape 256 256 256 --nte 8
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_exp9_slot_and_chunk.txt
