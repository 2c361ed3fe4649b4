# round 3: parity at scale beyond the low-field / eight-echo case of r03_parity_50k.sh -- the high-field tables and trains of
# six and three echoes, 30 000 voxels per configuration, all models: HIP vs the live oracle with the one-ulp yardstick beside it
cd $GRAFT_REPO_ROOT
show() {
python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], d["tables"], len(d["te_ms"]), "echoes", d["n_voxels"], "voxels")
for k, v in d["configs"].items():
    if "hip_lbfgsb_vs_reference" in v:
        h, r = v["hip_lbfgsb_vs_reference"], v["reference_vs_itself_one_ulp"]
        print(f"{k:38s} HIP within 1 ms {h['within_1ms']:.4f}  reference vs itself {r['within_1ms']:.4f}  success equal {h['success_equal']:.4f}  nit equal {h['nit_equal']:.4f}")
PY
}
timeout -k 10 900 python tools/parity_at_scale.py 30000 --all --hf > gpurun_out/r03_parity_at_scale_30k_hf.json 2> gpurun_out/r03_parity_hf.err && show gpurun_out/r03_parity_at_scale_30k_hf.json &&
timeout -k 10 900 python tools/parity_at_scale.py 30000 --all --n-te 6 > gpurun_out/r03_parity_at_scale_30k_te6.json 2> gpurun_out/r03_parity_te6.err && show gpurun_out/r03_parity_at_scale_30k_te6.json &&
timeout -k 10 900 python tools/parity_at_scale.py 30000 --all --n-te 3 > gpurun_out/r03_parity_at_scale_30k_te3.json 2> gpurun_out/r03_parity_te3.err && show gpurun_out/r03_parity_at_scale_30k_te3.json
