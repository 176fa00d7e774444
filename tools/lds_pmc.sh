export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --output-format csv -d $O/r04_pmc_batch1_lds -- python3 bench.py --only batch1 --no-graph --no-prime --steps 2 --warmup 1 --reps 0 --sync-steps > $O/r04_pmc_batch1_lds.log 2>&1 || { echo failed; tail -5 $O/r04_pmc_batch1_lds.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, os
f=max(glob.glob(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r04_pmc_batch1_lds/**/*counter_collection.csv'),recursive=True),key=os.path.getsize)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in sorted(acc.items(), key=lambda kv:-kv[1].get('SQ_LDS_IDX_ACTIVE',0))[:14]:
    print(k, {c:int(x) for c,x in v.items()})
PY
