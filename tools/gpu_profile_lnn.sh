#!/bin/bash
# Run on the GPU box: rocprofv3 kernel statistics of the whole-network step (tools/bench_lnn.py [--config kitti|shapenet|scannet]).
set -u
CONFIG=${1:-kitti}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_lnn
[ "$CONFIG" != "kitti" ] && OUT=$ROOT/gpurun_out/prof_lnn_$CONFIG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/tools/bench_lnn.py --config $CONFIG --steps 10 --warmup 3 > $OUT/stats.log 2>&1
cd $ROOT
tail -2 $OUT/stats.log
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total GPU time {tot/13/1e6:.2f} ms per step (13 steps), {sum(int(r['Calls']) for r in rows)/13:.0f} launches per step")
for r in rows[:28]:
    print(f'{r["Name"][:90]:90s} calls/step {int(r["Calls"])/13:7.1f}  avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]}%')
PY
