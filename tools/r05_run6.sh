#!/bin/bash
mkdir -p gpurun_out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --only batch8 --reps 3 "$@" > gpurun_out/r05_x_$tag.json 2> gpurun_out/r05_x_$tag.err || exit 1; }
run a_long --steps 150 --warmup 30
run b_short --steps 20 --warmup 5
run c_s20w30 --steps 20 --warmup 30
run d_s150w5 --steps 150 --warmup 5
run e_s30w30 --steps 30 --warmup 30
run f_s60w0 --steps 60 --warmup 0
run g_short --steps 20 --warmup 5
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_x_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('r05_x_')[1], round(d['value'],2), round(d['ms_per_step'],3), round(d['ms_per_step_median_of_5_passes'],3), d['kernel_class_ms_per_forward']['conv3x3'])
PY
