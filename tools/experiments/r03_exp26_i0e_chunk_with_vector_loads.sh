# round 3: coefficients per trip of the shared i0e loop (T2_I0E_CHUNK) re-measured with the per-lane vector loads of exp25:
# 3 / 5 (in-tree) / 6 / 10
cd $GRAFT_REPO_ROOT
D=$PWD/tools/diag
for args in "--fit rician --shape 180 256 256 --nte 6" "--fit rician"; do
for k in 3 6 10; do T2FIT_LIB=$D/libt2fit_k$k.so python tools/kernel_ab.py chunk_$k $args 2>/dev/null | tail -1 || exit 1; done
python tools/kernel_ab.py chunk_5 $args 2>/dev/null | tail -1
done
