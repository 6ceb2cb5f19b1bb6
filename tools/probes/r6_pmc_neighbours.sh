#!/bin/bash
# counters of k_neighbours / k_point_keys / k_bucket_rows under both slot orders (tools/probes/r6_kernels.py, REPS=4)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r6nb; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for so in hash space; do
  for set in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    REPS=4 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/x -o x -- python3 $ROOT/tools/probes/r6_kernels.py $so > $OUT/${so}_$tag.log 2>&1
    s=$(find $OUT/x -name "*counter_collection.csv" | head -1); cp "$s" $OUT/${so}_$tag.csv; rm -rf $OUT/x
  done
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, sys, glob, collections
for so in ("hash", "space"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{sys.argv[1]}/{so}_*.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
            if any(s in k for s in ("k_neighbours", "k_point_keys", "k_bucket_rows")):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(so, k, "  ".join(f"{c}={sum(v)/len(v):.3g}" for c, v in sorted(d.items())))
PY
