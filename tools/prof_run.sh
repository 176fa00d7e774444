#!/bin/bash
# Profiling passes of one round (run on the GPU box through gpurun; outputs under gpurun_out/, summarised into profiles/ by
# tools/prof_collect.py).  Usage: bash tools/prof_run.sh <tag, e.g. r02> [kernel|pmc|all]
TAG=${1:-r02}
WHAT=${2:-all}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
if [ "$WHAT" != "pmc" ]; then
  for W in batch8 batch1; do
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$W -- python3 bench.py --only $W --steps 60 --warmup 30 > $O/${TAG}_prof_$W.json 2> $O/${TAG}_prof_$W.err || { echo "kernel-trace $W failed"; exit 1; }
    echo "kernel-trace $W done"
  done
fi
if [ "$WHAT" != "kernel" ]; then
  # HBM-side traffic: one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass), then cache hits and SQ activity
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    n=$(echo $C | cut -d" " -f1)
    timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_b8_$n -- python3 tools/pmc_unet.py 8 > $O/${TAG}_pmc_b8_$n.log 2>&1 || { echo "pmc $n failed"; exit 1; }
    echo "pmc $n done"
  done
  python3 tools/prof_collect.py pmc batch8 $O/${TAG}_b8_pmc.csv $O/${TAG}_pmc_b8_*/
  cp profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json
  find $O -name "*_counter_collection.csv" -size +8M -delete
fi
find $O -name "*_kernel_trace.csv" -size +8M -delete
echo profiling done
