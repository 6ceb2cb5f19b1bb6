#!/bin/bash
# LDS counters of the filter-gradient kernels over tools/probes/gf_time.py (GPU box): instructions, bank-conflict cycles, issue stalls, LDS-array cycles
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_gf; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o a -- python3 $ROOT/tools/probes/gf_time.py > $OUT/a.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/a/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    if "grad_filter" in k or "backward_fused" in k: acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in acc: print(k[:60], {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()})
PY
