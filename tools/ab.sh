#!/bin/bash
# A/B harness for the GPU box: one rocprofv3 kernel-stats run of bench.py (graph mode) per configuration, compact table.
# Usage: bash tools/ab.sh "name1:ENV1=a ENV2=b" "name2:" ...     (env assignments apply to that run only)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/ab
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "$@"; do
  name=${cfg%%:*}
  envs=${cfg#*:}
  (
    for kv in $envs; do export "$kv"; done
    cd /tmp
    timeout ${AB_TIMEOUT:-90} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o s -- python3 $ROOT/bench.py --steps 30 --warmup 5 --cpu-seconds 0 --full-unet 0 --extra-kernels "" ${BENCH_ARGS:-} > $OUT/$name.log 2>&1
  )
  f=$(find $OUT/$name -name '*kernel_stats.csv' | head -1)
  cp "$f" $OUT/$name.kernel_stats.csv 2>/dev/null
  rm -rf $OUT/$name
done
python3 - $OUT "$@" <<'PY'
import csv, json, sys, os, re
out = sys.argv[1]
names = [c.split(":")[0] for c in sys.argv[2:]]
table, order = {}, []
for n in names:
    val = "-"
    try:
        for line in open(f"{out}/{n}.log"):
            if line.startswith("{"):
                d = json.loads(line); val = f"{d['ms_per_step']*1e3:.1f}"
    except Exception as e:
        pass
    table.setdefault("== us/step (bench wall)", {})[n] = val
    try:
        rows = list(csv.DictReader(open(f"{out}/{n}.kernel_stats.csv")))
    except Exception:
        continue
    hot = [r for r in rows if re.match(r"(void )?(k_|ln_k)", r["Name"])]
    if not hot: continue
    calls = max(int(r["Calls"]) for r in hot)
    tot = 0.0
    for r in hot:
        k = re.sub(r"\(.*", "", r["Name"].replace("void ", ""))[:44]
        per = float(r["TotalDurationNs"]) / calls / 1e3
        tot += per
        table.setdefault(k, {})[n] = f"{per:.2f}"
    table.setdefault("== sum kernels/step", {})[n] = f"{tot:.1f}"
print("kernel".ljust(46) + "".join(n[:12].rjust(13) for n in names))
for k in sorted(table, key=lambda k: (k.startswith("=="), k)):
    print(k.ljust(46) + "".join(table[k].get(n, "-").rjust(13) for n in names))
PY
