#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel statistics of bench.py — the default execution (four scans in flight) and
# one scan at a time —, of the splat -> slice pair alone (tools/chain_inflight.py), and the two HBM-traffic PMC passes.
# --extras 0: only capture / warm-up / timed replays + the eager steps that time the roofline group, so call counts are per step.
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="--steps 300 --warmup 10 --cpu-seconds 0 --full-unet 0 --extras 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py $B > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o stats1 -- python3 $ROOT/bench.py $B --in-flight 1 > $OUT/stats1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain -o chain -- python3 $ROOT/tools/chain_inflight.py --in-flight 3 --prefetch 0 --reps 300 > $OUT/chain.log 2>&1
P="--steps 8 --warmup 2 --cpu-seconds 0 --full-unet 0 --extras 0 --in-flight 1"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py $P > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $ROOT/bench.py $P > $OUT/write.log 2>&1
cd $ROOT
for t in stats stats1 chain; do
  S=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); cp "$S" $OUT/${t}_kernel_stats.csv 2>/dev/null
done
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); cp "$F" $OUT/pmc_fetch_size_counter_collection.csv
W=$(find $OUT/write -name "*counter_collection.csv" | head -1); cp "$W" $OUT/pmc_write_size_counter_collection.csv
python3 tools/pmc_traffic.py $OUT/pmc_fetch_size_counter_collection.csv $OUT/pmc_write_size_counter_collection.csv $OUT/pmc_traffic.json > /dev/null
grep -h "^{\"metric\"" $OUT/stats.log > $OUT/bench_line_in_flight_default.json
grep -h "^{\"metric\"" $OUT/stats1.log > $OUT/bench_line_in_flight1.json
rm -rf $OUT/stats $OUT/stats1 $OUT/chain $OUT/fetch $OUT/write
for t in stats stats1 chain; do echo "== $t"; head -14 $OUT/${t}_kernel_stats.csv | cut -c1-150; done
python3 - $OUT <<'PY'
import json, sys
t = json.load(open(sys.argv[1] + "/pmc_traffic.json"))
for k, v in t.items():
    if isinstance(v, dict): print(f"{k:32s} fetch {v['fetch_size_kb_raw']*2/1024:8.1f} MB  write {v['write_size_kb']/1024:8.1f} MB  total {v['traffic_bytes']/1e6:8.1f} MB")
PY
