#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r05_t4.log 2>&1
rc=$?
tail -5 gpurun_out/r05_t4.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python tools/ab_unet.py 16 1 2 > gpurun_out/r05_ab_fork.txt 2>&1 || exit 1
cat gpurun_out/r05_ab_fork.txt
timeout -k 10 300 python tools/ab_unet.py 524288 1 > gpurun_out/r05_ab_2wg.txt 2>&1 || exit 1
cat gpurun_out/r05_ab_2wg.txt
for i in 1 2; do
LD_MI355X_LIB=lightdiffusion_amd/libld_r04.so timeout -k 10 300 python tools/unet_time.py 1 8 >> gpurun_out/r05_time4.txt 2>&1 || exit 1
timeout -k 10 300 python tools/unet_time.py 1 8 >> gpurun_out/r05_time4.txt 2>&1 || exit 1
done
grep median gpurun_out/r05_time4.txt
exit $rc
