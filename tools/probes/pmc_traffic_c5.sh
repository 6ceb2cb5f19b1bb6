#!/bin/bash
# HBM-side traffic per launch of the C5 step's kernels (FETCH_SIZE / WRITE_SIZE passes, one in flight): do the segment reduces re-read their
# point rows from beyond the L2s?
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_c5; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
P="--workload C5 --steps 8 --warmup 2 --cpu-seconds 0 --full-unet 0 --extras 0 --in-flight 1"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o $c -- python3 $ROOT/bench.py $P > $OUT/$c.log 2>&1
  cp $(find $OUT/$c -name "*counter_collection.csv" | head -1) $OUT/${c}_counter_collection.csv; rm -rf $OUT/$c
done
cd $ROOT
python3 tools/pmc_traffic.py $OUT/FETCH_SIZE_counter_collection.csv $OUT/WRITE_SIZE_counter_collection.csv $OUT/pmc_traffic_c5.json | head -40
