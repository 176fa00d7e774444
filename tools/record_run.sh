#!/bin/bash
# record run of a round (r05, r06, ...): full -m gpu suite, bench lines (default and driver-style), launch tables.  Usage (gpurun): bash tools/record_run.sh <tag>
TAG=${1:-r05}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/${TAG}_gpu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/${TAG}_gpu_tests.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
cp gpurun_out/e2e_parity.json gpurun_out/${TAG}_e2e_parity.json 2>/dev/null
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_steps20_warmup5.json 2> gpurun_out/${TAG}_bench_steps20_warmup5.err || exit 1
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
grep -h "steps in" gpurun_out/${TAG}_bench_steps20_warmup5.err gpurun_out/${TAG}_bench.err
timeout -k 10 300 python tools/ab_launches.py 8 0 > gpurun_out/${TAG}_launches_b8.txt 2>&1 || exit 1
timeout -k 10 300 python tools/ab_launches.py 1 0 > gpurun_out/${TAG}_launches_b1.txt 2>&1 || exit 1
timeout -k 10 300 python tools/launch_table.py 8 1 > gpurun_out/${TAG}_launch_table_b8_pair.txt 2>&1 || exit 1
timeout -k 10 300 python tools/launch_table.py 1 1 > gpurun_out/${TAG}_launch_table_b1_pair.txt 2>&1 || exit 1
exit $rc
