#!/bin/bash
mkdir -p gpurun_out
LD_MI355X_LIB=lightdiffusion_amd/libld_r04.so timeout -k 10 300 python tools/launch_table.py 1 > gpurun_out/r05_lt_b1_r04.txt 2>&1 || exit 1
timeout -k 10 300 python tools/launch_table.py 1 > gpurun_out/r05_lt_b1_new.txt 2>&1 || exit 1
timeout -k 10 300 python tools/ab_launches.py 1 0 2048 > gpurun_out/r05_launches_b1_ring.txt 2>&1 || exit 1
timeout -k 10 300 python tools/ab_launches.py 8 0 1024 > gpurun_out/r05_launches_b8_geglu5.txt 2>&1 || exit 1
