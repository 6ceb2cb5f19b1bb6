#!/bin/bash
# Run on the GPU box: rocprofv3 kernel statistics of the hot-path chain only (no PMC passes); prints the per-kernel averages.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/quick_prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --full-unet 0 "$@" > $OUT/stats.log 2>&1
cd $ROOT
grep "^{" $OUT/stats.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench', d['value'], 'Mpts/s', d['ms_per_step'], 'ms')"
python3 - "$(find $OUT/stats -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
hot = [r for r in rows if r["Name"].startswith(("void k_", "k_", "void ln_k"))]
tot = 0.0
calls = max(int(r["Calls"]) for r in hot)
for r in hot:
    per_step = float(r["TotalDurationNs"]) / calls / 1e3
    tot += per_step
    print(f'{r["Name"][:64]:64s} calls {int(r["Calls"]):4d}  avg {float(r["AverageNs"])/1e3:7.2f} us')
print(f"sum per step {tot:.1f} us")
PY
