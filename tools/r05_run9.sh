#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_models_gpu.py tests/test_configs_gpu.py tests/test_conv8_gpu.py tests/test_edges_gpu.py tests/test_properties_gpu.py -m gpu -q --timeout 600 > gpurun_out/r05_t9.log 2>&1
rc=$?
tail -5 gpurun_out/r05_t9.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
rm -f gpurun_out/r05_vae_time.txt
for i in 1 2; do
LD_MI355X_LIB=lightdiffusion_amd/libld_r04.so timeout -k 10 300 python tools/vae_time.py >> gpurun_out/r05_vae_time.txt 2>&1 || exit 1
timeout -k 10 300 python tools/vae_time.py >> gpurun_out/r05_vae_time.txt 2>&1 || exit 1
done
cat gpurun_out/r05_vae_time.txt
exit $rc
