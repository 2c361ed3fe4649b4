# BASELINE.json configs that fit one GPU, one bench line each (kernel ms / Mvoxel/s); not the headline.
run() { python bench.py "$@" --steps 5 --warmup 1 --cpu-seconds 0 --no-also 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(json.dumps({'workload': d['config']['workload'], 'Mvoxel_s': d['value'], 'kernel_ms': d['roofline']['kernel_ms'], 'step_ms': d['ms_per_step'], 'hbm_frac': d['roofline']['frac']}))"; }
run --shape 20 64 64 --n-te 6 --fit gaussian --no-prior                  # cfg1 phantom size
run --shape 180 256 256 --n-te 6 --fit gaussian                          # cfg2 (2-parameter)
run --shape 180 256 256 --n-te 6 --fit gaussian --solver lm --precision f32
run --shape 180 256 256 --n-te 6 --fit gaussian --solver loglin          # cfg2 as BASELINE.json words it: log-linear closed form
run --shape 180 256 256 --n-te 6 --fit gaussian_rician                   # cfg3 (3-parameter)
run --shape 180 256 256 --n-te 6 --fit gaussian_rician --solver lm --precision f32
run --shape 180 256 256 --n-te 6 --fit rician                            # Rician likelihood model
run --shape 45 512 512 --n-te 8 --fit gaussian_rician                    # cfg4: one of 8 slabs of 512x512x360
run --shape 256 256 256 --n-te 8 --fit gaussian_rician --no-prior
