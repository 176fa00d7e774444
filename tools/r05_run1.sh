#!/bin/bash
# round-5 GPU call 1: full -m gpu suite, then same-box timings (new library vs round 4's) and the skip-fold A/B
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r05_t1.log 2>&1
rc=$?
tail -5 gpurun_out/r05_t1.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
LD_MI355X_LIB=lightdiffusion_amd/libld_r04.so timeout -k 10 300 python tools/unet_time.py 8 1 > gpurun_out/r05_time_r04.txt 2>&1 || exit 1
timeout -k 10 300 python tools/unet_time.py 8 1 > gpurun_out/r05_time_new.txt 2>&1 || exit 1
cat gpurun_out/r05_time_r04.txt gpurun_out/r05_time_new.txt
timeout -k 10 300 python tools/ab_unet.py 8 8 > gpurun_out/r05_ab_skip.txt 2>&1 || exit 1
cat gpurun_out/r05_ab_skip.txt
timeout -k 10 300 python tools/ab_launches.py 8 0 > gpurun_out/r05_launches_b8_a.txt 2>&1 || exit 1
tail -3 gpurun_out/r05_launches_b8_a.txt
exit $rc
