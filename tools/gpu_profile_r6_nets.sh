#!/bin/bash
# Round 6: kernel statistics of the workloads the verdict found unprofiled — the ScanNet network step, the C4 chain (one scan at a time),
# the ShapeNet network step and the C2 chain.  Outputs: gpurun_out/prof_r6_nets/*_kernel_stats.csv (+ a printed top list).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r6_nets; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
stats() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o x -- "$@" > $OUT/$name.log 2>&1
  local s=$(find $OUT/$name -name "*kernel_stats.csv" | head -1); cp "$s" $OUT/${name}_kernel_stats.csv 2>/dev/null; rm -rf $OUT/$name
  tail -2 $OUT/$name.log | cut -c1-250
}
for w in ${WHAT:-lnn_scannet c4 lnn_shapenet c2}; do
  case $w in
    lnn_scannet) stats lnn_scannet python3 $ROOT/tools/bench_lnn.py --config scannet --steps 6 --warmup 2 ;;
    lnn_shapenet) stats lnn_shapenet python3 $ROOT/tools/bench_lnn.py --config shapenet --steps 10 --warmup 3 ;;
    c4) stats c4 python3 $ROOT/bench.py --workload C4 --steps 100 --warmup 5 --cpu-seconds 0 --in-flight 1 --extras 0 ;;
    c2) stats c2 python3 $ROOT/bench.py --workload C2 --steps 300 --warmup 10 --cpu-seconds 0 --in-flight 1 --extras 0 ;;
  esac
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*_kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print("==", os.path.basename(f), f"total {tot/1e6:.2f} ms, {sum(int(r['Calls']) for r in rows)} launches")
    for r in rows[:22]:
        print(f'   {r["Name"][:86]:86s} calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]}%')
PY
