#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_configs_gpu.py -m gpu -q --timeout 600 > gpurun_out/r05_t10.log 2>&1
rc=$?
tail -5 gpurun_out/r05_t10.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python tools/launch_table.py 8 1 > gpurun_out/r05_lt_b8_new2.txt 2>&1 || exit 1
head -40 gpurun_out/r05_lt_b8_new2.txt
rm -f gpurun_out/r05_time10.txt
for i in 1 2; do
LD_MI355X_LIB=lightdiffusion_amd/libld_r04.so timeout -k 10 300 python tools/unet_time.py 8 >> gpurun_out/r05_time10.txt 2>&1 || exit 1
timeout -k 10 300 python tools/unet_time.py 8 >> gpurun_out/r05_time10.txt 2>&1 || exit 1
done
grep median gpurun_out/r05_time10.txt
exit $rc
