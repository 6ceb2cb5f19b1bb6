#!/bin/bash
# Which fill launches does one whole-network step issue: grid sizes of the FillFunctor kernels from a rocprofv3 kernel trace.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/fill_sizes; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 $ROOT/tools/bench_lnn.py --config kitti --steps 10 --warmup 3 "$@" > $OUT/log.txt 2>&1
cd $ROOT
python3 - $(find $OUT/t -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
c = collections.Counter()
for r in rows:
    if "FillFunctor" in r["Kernel_Name"]:
        ty = r["Kernel_Name"].split("FillFunctor<")[1].split(">")[0]
        c[(ty, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))] += 1
for (ty, g), n in sorted(c.items(), key=lambda kv: -kv[1]):
    print(f"{n/13:6.1f}/step  FillFunctor<{ty}> grid {g}")
PY
rm -rf $OUT/t
