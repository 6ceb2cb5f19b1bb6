#!/bin/bash
# LDS counters of every kernel of the C3 step and the full network step (GPU box): instructions, bank-conflict cycles, LDS issue stalls, LDS-array cycles
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_lds_unet; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -o a -- python3 $ROOT/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --full-unet 1 --extras 0 --in-flight 1 > $OUT/a.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/a/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    if k.startswith("void k_") or k.startswith("k_"): acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in acc: print(k[:60].ljust(60), {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()})
PY
