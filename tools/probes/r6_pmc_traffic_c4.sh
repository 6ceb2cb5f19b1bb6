#!/bin/bash
# HBM traffic of the C4 chain's kernels (verdict item 5): separate FETCH_SIZE / WRITE_SIZE passes, kernel trace only
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_c4; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
P="--workload C4 --steps 8 --warmup 2 --cpu-seconds 0 --extras 0 --in-flight 1"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o x -- python3 $ROOT/bench.py $P > $OUT/$c.log 2>&1
  cp "$(find $OUT/$c -name '*counter_collection.csv' | head -1)" $OUT/${c}_counter_collection.csv
done
python3 $ROOT/tools/pmc_traffic.py $OUT/FETCH_SIZE_counter_collection.csv $OUT/WRITE_SIZE_counter_collection.csv $OUT/pmc_traffic_C4.json
python3 - <<PY
import json
t=json.load(open("$OUT/pmc_traffic_C4.json"))
print({k: round(v["traffic_bytes"]/1e6,1) for k,v in t.items() if isinstance(v,dict)}, "sum", round(sum(v["traffic_bytes"] for v in t.values() if isinstance(v,dict))/1e6,1), "MB")
PY
