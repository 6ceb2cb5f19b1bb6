#!/bin/bash
# Which kernels of a whole-network step run with few workgroups for how long (candidates for "one wave per SIMD" latency problems).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/grid_census; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 $ROOT/tools/bench_lnn.py --config kitti --steps 10 --warmup 3 > $OUT/log.txt 2>&1
cd $ROOT
python3 - $(find $OUT/t -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
    gx = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0); wx = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1)
    gy = int(r.get("Grid_Size_Y") or 1); gz = int(r.get("Grid_Size_Z") or 1); wy = int(r.get("Workgroup_Size_Y") or 1); wz = int(r.get("Workgroup_Size_Z") or 1)
    wgs = (gx // max(wx, 1)) * (gy // max(wy, 1)) * (gz // max(wz, 1))
    waves = wgs * ((wx * wy * wz + 63) // 64)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = (name, wgs, waves)
    agg[k][0] += 1; agg[k][1] += d
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f"{'kernel':70s} {'wgs':>7s} {'waves':>7s} {'n/step':>7s} {'avg us':>8s} {'us/step':>8s}")
for (name, wgs, waves), (n, tot, _) in out[:70]:
    if waves <= 4096 and tot / n >= 8.0:
        print(f"{name:70s} {wgs:7d} {waves:7d} {n/13:7.1f} {tot/n:8.1f} {tot/13:8.1f}")
PY
rm -rf $OUT/t
