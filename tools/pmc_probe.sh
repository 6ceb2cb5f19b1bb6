#!/bin/bash
# Run on the GPU box: a few rocprofv3 counter passes over the hot-path chain (each pass = its own run, counters only with
# --kernel-trace), then per-kernel averages.  Usage: bash tools/pmc_probe.sh "CTR_A CTR_B" "CTR_C" ...
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_probe
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for group in "$@"; do
    i=$((i + 1))
    timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $ROOT/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --full-unet 0 > $OUT/p$i.log 2>&1
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith(("k_", "ln_k")):
            continue
        acc[name.split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
ctrs = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(28) + "".join(c[-22:].rjust(24) for c in ctrs))
for k in sorted(acc):
    print(k.ljust(28) + "".join((f"{sum(acc[k][c]) / len(acc[k][c]):.4g}" if acc[k][c] else "-").rjust(24) for c in ctrs))
PY
