#!/bin/bash
# LDS counters of every kernel of a network step (GPU box): bank-conflict cycles against all LDS-array cycles.  CONFIG=kitti|scannet|shapenet
set -u
CONFIG=${CONFIG:-kitti}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_lds_$CONFIG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -o a -- python3 $ROOT/tools/bench_lnn.py --config $CONFIG --steps 3 --warmup 1 > $OUT/a.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/a/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
rows=[]
for k in acc:
    d={c: v/cnt[(k,c)] for c,v in acc[k].items()}
    if d.get("SQ_LDS_IDX_ACTIVE",0)>0: rows.append((d["SQ_LDS_BANK_CONFLICT"]*cnt[(k,"SQ_LDS_BANK_CONFLICT")],k,d,cnt[(k,"SQ_LDS_BANK_CONFLICT")]))
print("kernel | launches | LDS-array cycles per launch | conflict cycles per launch | share | busy cycles per launch")
for tot,k,d,n in sorted(rows,reverse=True):
    print(f'{k[:70]:70s} {n:5d} {d["SQ_LDS_IDX_ACTIVE"]:11.0f} {d["SQ_LDS_BANK_CONFLICT"]:11.0f} {100*d["SQ_LDS_BANK_CONFLICT"]/d["SQ_LDS_IDX_ACTIVE"]:5.1f}% {d["SQ_BUSY_CYCLES"]:11.0f}')
PY
