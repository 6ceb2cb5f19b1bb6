#!/bin/bash
# FETCH_SIZE / L2 hit-rate per kernel for a list of configurations "name:ENV=..".  One scan in flight.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_ab
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for ctr in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $ctr | cut -c1-8 | tr ' ' '_')
    ( for kv in $envs; do export "$kv"; done
      cd /tmp
      timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${name}_$tag -o p -- python3 $ROOT/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --full-unet 0 --in-flight 1 --extra-kernels "" > $OUT/${name}_$tag.log 2>&1 )
    f=$(find $OUT/${name}_$tag -name '*counter_collection.csv' | head -1)
    cp "$f" $OUT/${name}_$tag.csv 2>/dev/null; rm -rf $OUT/${name}_$tag
  done
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/*.csv")):
    cfg = os.path.basename(f).rsplit("_", 1)[0]
    for r in csv.DictReader(open(f)):
        m = re.match(r"(?:void )?(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m: acc[(m.group(1), cfg)][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':28s}{'config':12s}{'fetch MB(x2)':>14s}{'L2 hit %':>10s}")
for (k, cfg) in sorted(acc):
    a = acc[(k, cfg)]
    fe = sum(a["FETCH_SIZE"]) / max(len(a["FETCH_SIZE"]), 1) * 2 / 1024 if a["FETCH_SIZE"] else float("nan")
    h = sum(a["TCC_HIT_sum"]); mi = sum(a["TCC_MISS_sum"])
    print(f"{k:28s}{cfg:12s}{fe:14.1f}{(100 * h / (h + mi) if h + mi else float('nan')):10.1f}")
PY
