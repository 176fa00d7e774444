#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r05_t8.log 2>&1
rc=$?
tail -5 gpurun_out/r05_t8.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
rm -f gpurun_out/r05_time8.txt
for i in 1 2; do
LD_MI355X_LIB=lightdiffusion_amd/libld_r04.so timeout -k 10 300 python tools/unet_time.py 1 8 >> gpurun_out/r05_time8.txt 2>&1 || exit 1
timeout -k 10 300 python tools/unet_time.py 1 8 >> gpurun_out/r05_time8.txt 2>&1 || exit 1
done
grep median gpurun_out/r05_time8.txt
timeout -k 10 300 python tools/launch_table.py 8 1 > gpurun_out/r05_lt_b8_new.txt 2>&1 || exit 1
LD_MI355X_LIB=lightdiffusion_amd/libld_r04.so timeout -k 10 300 python tools/launch_table.py 8 1 > gpurun_out/r05_lt_b8_r04.txt 2>&1 || exit 1
exit $rc
