#!/bin/bash
# workgroup sizes of the chain's kernels as dispatched by bench.py (kernel trace)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/wg; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/a -o a -- python3 $ROOT/bench.py ${BENCH_ARGS:-} --steps 40 --warmup 5 --cpu-seconds 0 --full-unet 0 --extras 0 > $OUT/a.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/a/**/*kernel_trace.csv",recursive=True)[0]
c=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    if "k_" in k: c[(k[:50], r.get("Workgroup_Size_X") or r.get("Workgroup_Size"), r.get("Grid_Size_X") or r.get("Grid_Size"))]+=1
for k,v in sorted(c.items()): print(k,v)
PY
