#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_configs_gpu.py tests/test_conv8_gpu.py -m gpu -q --timeout 600 > gpurun_out/r05_t7.log 2>&1
rc=$?
tail -5 gpurun_out/r05_t7.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python tools/gemm3_abl.py quick > gpurun_out/r05_gemm3_abl3.txt 2>&1 || exit 1
cat gpurun_out/r05_gemm3_abl3.txt
for i in 1 2; do
LD_MI355X_LIB=lightdiffusion_amd/libld_r04.so timeout -k 10 300 python tools/unet_time.py 1 8 >> gpurun_out/r05_time7.txt 2>&1 || exit 1
timeout -k 10 300 python tools/unet_time.py 1 8 >> gpurun_out/r05_time7.txt 2>&1 || exit 1
done
grep median gpurun_out/r05_time7.txt
exit $rc
