#!/bin/bash
# Profiling passes of one round (run on the GPU box through gpurun; outputs under gpurun_out/, summarised into profiles/ by
# tools/prof_collect.py).  Usage: bash tools/prof_run.sh <tag, e.g. r03> [kernel|pmc|all] [workloads, default "batch8 batch1 hires"]
# Every pass drives bench.py itself (python3 directly after `--`).  The --pmc passes run it eagerly with a synchronize per sampler
# step (--no-graph --sync-steps --reps 0): rocprofiler-sdk's AQL write interceptor reads past its 1 MiB packet buffer when the
# application has thousands of launches queued (profiles/README.md, round 3), and counter collection serialises every dispatch
# (~10 ms each), so a pass is 3 steps + the 3 event-profiled forwards.
TAG=${1:-r03}
WHAT=${2:-all}
LOADS=${3:-"batch8 batch1 hires"}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
if [ "$WHAT" != "pmc" ]; then
  for W in $LOADS; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$W -- python3 bench.py --only $W --steps 60 --warmup 30 > $O/${TAG}_prof_$W.json 2> $O/${TAG}_prof_$W.err || { echo "kernel-trace $W failed"; tail -3 $O/${TAG}_prof_$W.err; exit 1; }
    python3 tools/prof_collect.py stats $O/${TAG}_prof_$W profiles/${TAG}_${W}_kernel_stats.csv && cp profiles/${TAG}_${W}_kernel_stats.csv $O/
    echo "kernel-trace $W done"
  done
  for V in "8 64 vae512" "4 128 vae1024"; do
    set -- $V
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$3 -- python3 tools/vae_prof.py $1 $2 > $O/${TAG}_$3_launches.txt 2> $O/${TAG}_prof_$3.err || { echo "kernel-trace $3 failed"; exit 1; }
    python3 tools/prof_collect.py stats $O/${TAG}_prof_$3 profiles/${TAG}_$3_kernel_stats.csv && cp profiles/${TAG}_$3_kernel_stats.csv $O/ && cp $O/${TAG}_$3_launches.txt profiles/
    echo "kernel-trace $3 done"
  done
fi
if [ "$WHAT" != "kernel" ]; then
  # HBM-side traffic: one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass), then cache hits and SQ activity
  for W in $LOADS; do
    SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum")
    [ "$W" = "batch8" ] && SETS+=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE")
    for C in "${SETS[@]}"; do
      n=$(echo $C | cut -d" " -f1)
      timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_${W}_$n -- python3 bench.py --only $W --no-graph --no-prime --steps 2 --warmup 1 --reps 0 --sync-steps > $O/${TAG}_pmc_${W}_$n.log 2>&1 || { echo "pmc $W $n failed"; tail -3 $O/${TAG}_pmc_${W}_$n.log; exit 1; }
      echo "pmc $W $n done"
    done
    if [ "$W" = "batch8" ]; then
      # co-execution / wait / LDS-instruction counters (round 4; a counter this rocprofiler build does not know only loses this pass)
      C="SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
      timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_${W}_SQ_VALU_MFMA_COEXEC_CYCLES -- python3 bench.py --only $W --no-graph --no-prime --steps 2 --warmup 1 --reps 0 --sync-steps > $O/${TAG}_pmc_${W}_coexec.log 2>&1 && echo "pmc $W coexec done" || { echo "pmc $W coexec pass failed (kept going)"; tail -3 $O/${TAG}_pmc_${W}_coexec.log; rm -rf $O/${TAG}_pmc_${W}_SQ_VALU_MFMA_COEXEC_CYCLES; }
    fi
    python3 tools/prof_collect.py pmc $W $O/${TAG}_${W}_pmc.csv $O/${TAG}_pmc_${W}_*/ && cp $O/${TAG}_${W}_pmc.csv profiles/
  done
  for V in "8 64 vae512"; do
    set -- $V
    for C in "FETCH_SIZE" "WRITE_SIZE"; do
      timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_$3_$C -- python3 tools/vae_prof.py $1 $2 > $O/${TAG}_pmc_$3_$C.log 2>&1 || { echo "pmc $3 $C failed"; exit 1; }
      echo "pmc $3 $C done"
    done
    LD_PROF_DRIVER=tools/vae_prof.py python3 tools/prof_collect.py pmc $3 $O/${TAG}_$3_pmc.csv $O/${TAG}_pmc_$3_*/ && cp $O/${TAG}_$3_pmc.csv profiles/
  done
  cp profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json
  find $O -name "*_counter_collection.csv" -size +8M -delete
fi
find $O -name "*_kernel_trace.csv" -size +8M -delete
echo profiling done
