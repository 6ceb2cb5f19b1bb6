#!/bin/bash
# where the cycles of the C3 chain's kernels go: VALU / MFMA / LDS / VMEM activity counters (separate --pmc passes), one scan at a time
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6conv; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
P="--steps 8 --warmup 2 --cpu-seconds 0 --full-unet 0 --extras 0 --in-flight 1"
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/x -o x -- python3 $ROOT/bench.py $P > $OUT/set$i.log 2>&1
  s=$(find $OUT/x -name "*counter_collection.csv" | head -1); cp "$s" $OUT/set$i.csv 2>/dev/null; rm -rf $OUT/x
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/set*.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    busy = m.get("SQ_BUSY_CYCLES", 1)
    print(k)
    print("   " + "  ".join(f"{c.replace('SQ_', '')}={v:.3g}" for c, v in sorted(m.items())))
PY
