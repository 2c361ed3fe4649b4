# round 3: parity at scale, 50 000 voxels per configuration, all configurations incl. the rician objective as numpy 1.26
# evaluates it (cfg.numpy_legacy): HIP vs the live oracle on the box's host cores, with the one-ulp yardstick beside it
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python tools/parity_at_scale.py 50000 --all > gpurun_out/r03_parity_at_scale_50k_all.json 2> gpurun_out/r03_parity_at_scale.err
tail -3 gpurun_out/r03_parity_at_scale.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_parity_at_scale_50k_all.json"))
for k, v in d["configs"].items():
    if "hip_lbfgsb_vs_reference" in v:
        h, r = v["hip_lbfgsb_vs_reference"], v["reference_vs_itself_one_ulp"]
        print(f"{k:38s} HIP within 1 ms {h['within_1ms']:.4f}  reference vs itself {r['within_1ms']:.4f}  success equal {h['success_equal']:.4f}  nit equal {h['nit_equal']:.4f}")
PY
