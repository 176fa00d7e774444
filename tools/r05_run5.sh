#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --only batch8 --reps 3 > gpurun_out/r05_b8_short_$i.json 2> gpurun_out/r05_b8_short_$i.err || exit 1
LD_BENCH_PRIME_PASSES=5 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --only batch8 --reps 3 > gpurun_out/r05_b8_short_prime5_$i.json 2> gpurun_out/r05_b8_short_prime5_$i.err || exit 1
done
timeout -k 10 300 python bench.py --steps 150 --warmup 30 --only batch8 --reps 3 > gpurun_out/r05_b8_long.json 2> gpurun_out/r05_b8_long.err || exit 1
grep -h "steps in" gpurun_out/r05_b8_*.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_b8_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value'],2), round(d['ms_per_step'],3), d['ms_per_step_median_of_5_passes'], d.get('ms_per_step_passes'))
PY
