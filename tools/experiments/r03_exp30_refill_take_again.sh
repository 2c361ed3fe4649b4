# round 3: refill batch (T2FIT_REFILL_MIN) and chunks per counter increment (T2FIT_TAKE) re-measured on the final kernels
# (defaults: 8 / 2; Rician lane 4 / 2; the 2-parameter lane now runs ten waves per CU)
cd $GRAFT_REPO_ROOT
for r in 4 6 8 12; do T2FIT_REFILL_MIN=$r python tools/kernel_ab.py refill_$r 2>/dev/null | tail -1 || exit 1; done
for t in 1 2 3; do T2FIT_TAKE=$t python tools/kernel_ab.py take_$t 2>/dev/null | tail -1 || exit 1; done
for r in 4 8 12; do T2FIT_REFILL_MIN=$r python tools/kernel_ab.py refill_$r --fit gaussian --shape 180 256 256 --nte 6 2>/dev/null | tail -1 || exit 1; done
for r in 2 4 8; do T2FIT_REFILL_MIN=$r python tools/kernel_ab.py refill_$r --fit rician --shape 180 256 256 --nte 6 2>/dev/null | tail -1 || exit 1; done
