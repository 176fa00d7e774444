#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_configs_gpu.py -m gpu -q --timeout 600 > gpurun_out/r05_t2.log 2>&1
rc=$?
tail -5 gpurun_out/r05_t2.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python tools/ab_unet.py 524288 1 8 > gpurun_out/r05_ab_ring.txt 2>&1 || exit 1
cat gpurun_out/r05_ab_ring.txt
timeout -k 10 300 python tools/ab_unet.py 262144 8 > gpurun_out/r05_ab_geglu5.txt 2>&1 || exit 1
cat gpurun_out/r05_ab_geglu5.txt
for i in 1 2; do
LD_MI355X_LIB=lightdiffusion_amd/libld_r04.so timeout -k 10 300 python tools/unet_time.py 1 8 >> gpurun_out/r05_time2.txt 2>&1 || exit 1
timeout -k 10 300 python tools/unet_time.py 1 8 >> gpurun_out/r05_time2.txt 2>&1 || exit 1
done
grep median gpurun_out/r05_time2.txt
timeout -k 10 300 python tools/ab_launches.py 1 0 524288 > gpurun_out/r05_launches_b1_ring.txt 2>&1 || exit 1
exit $rc
