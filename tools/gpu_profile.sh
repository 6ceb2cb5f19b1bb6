#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + the two HBM-traffic PMC passes of bench.py.
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --full-unet 0 > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --full-unet 0 --in-flight 1 > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $ROOT/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --full-unet 0 --in-flight 1 > $OUT/write.log 2>&1
cd $ROOT
find $OUT -name "*.csv" | head -20
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W $OUT/pmc_traffic.json > /dev/null
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
head -25 $S | cut -c1-160
