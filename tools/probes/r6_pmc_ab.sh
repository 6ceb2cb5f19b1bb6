#!/bin/bash
# Round 6: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) and kernel statistics of the C3 step under both slot orders.
# Outputs: gpurun_out/r6pmc/{space,hash}_pmc_traffic.json, *_kernel_stats.csv
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r6pmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P="--steps 8 --warmup 2 --cpu-seconds 0 --full-unet 0 --extras 0 --in-flight 1"
B="--steps 300 --warmup 10 --cpu-seconds 0 --full-unet 0 --extras 0"
for so in ${ORDERS:-space hash}; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${so}_$ctr -o x -- python3 $ROOT/bench.py $P --slot-order $so > $OUT/${so}_$ctr.log 2>&1
    s=$(find $OUT/${so}_$ctr -name "*counter_collection.csv" | head -1); cp "$s" $OUT/${so}_${ctr}_counter_collection.csv; rm -rf $OUT/${so}_$ctr
  done
  python3 $ROOT/tools/pmc_traffic.py $OUT/${so}_FETCH_SIZE_counter_collection.csv $OUT/${so}_WRITE_SIZE_counter_collection.csv $OUT/${so}_pmc_traffic.json > /dev/null
  for fl in 1 4; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${so}_stats$fl -o x -- python3 $ROOT/bench.py $B --in-flight $fl --slot-order $so > $OUT/${so}_stats$fl.log 2>&1
    s=$(find $OUT/${so}_stats$fl -name "*kernel_stats.csv" | head -1); cp "$s" $OUT/${so}_in_flight${fl}_kernel_stats.csv; rm -rf $OUT/${so}_stats$fl
    grep -h "^{\"metric\"" $OUT/${so}_stats$fl.log | tail -1 | cut -c1-200
  done
done
cd $ROOT
python3 - $OUT <<'PY'
import json, sys, csv
for so in ("space", "hash"):
    try:
        t = json.load(open(f"{sys.argv[1]}/{so}_pmc_traffic.json"))
    except OSError:
        continue
    tot = 0
    print("==", so)
    for k, v in t.items():
        if isinstance(v, dict):
            tot += v["traffic_bytes"]
            print(f"{k:32s} fetch {v['fetch_size_kb_raw']*2/1024:8.1f} MB  write {v['write_size_kb']/1024:8.1f} MB  total {v['traffic_bytes']/1e6:8.1f} MB")
    print("sum", round(tot / 1e6, 1), "MB")
    for fl in (1, 4):
        try:
            rows = list(csv.DictReader(open(f"{sys.argv[1]}/{so}_in_flight{fl}_kernel_stats.csv")))
        except OSError:
            continue
        print(f"-- {so} in flight {fl}")
        for r in rows[:10]:
            print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']}%")
PY
