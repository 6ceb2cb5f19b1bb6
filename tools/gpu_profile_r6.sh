#!/bin/bash
# Round-6 profile set (run on the GPU box through gpurun; outputs in gpurun_out/prof_r6/, copied into profiles/r6_* by hand):
#   C3 chain under both slot orders: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes with --kernel-trace only) and
#   rocprofv3 kernel statistics one scan at a time and four in flight;
#   kernel statistics of the C4 / C2 chains (one scan at a time) and of the SemanticKITTI / ScanNet / ShapeNet network steps;
#   the bench lines (unprofiled) of C3 (default and the driver's command), C2, C4, C5;
#   LDS counters (instructions, bank-conflict cycles, LDS-array cycles) of the C3 step's kernels and of the SemanticKITTI network step.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r6; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
P="--steps 8 --warmup 2 --cpu-seconds 0 --full-unet 0 --extras 0 --in-flight 1"
B="--steps 300 --warmup 10 --cpu-seconds 0 --full-unet 0 --extras 0"
stats() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o x -- "$@" > $OUT/$name.log 2>&1
  local s=$(find $OUT/$name -name "*kernel_stats.csv" | head -1); cp "$s" $OUT/${name}_kernel_stats.csv 2>/dev/null; rm -rf $OUT/$name
  grep -h "^{\"metric\"" $OUT/$name.log | tail -1 > $OUT/${name}_bench_line.json 2>/dev/null
}
pmc() {
  local name=$1; local ctr=$2; shift; shift
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$name -o x -- "$@" > $OUT/$name.log 2>&1
  local s=$(find $OUT/$name -name "*counter_collection.csv" | head -1); cp "$s" $OUT/${name}_counter_collection.csv 2>/dev/null; rm -rf $OUT/$name
}
for so in hash space; do
  pmc pmc_fetch_$so FETCH_SIZE python3 $ROOT/bench.py $P --slot-order $so
  pmc pmc_write_$so WRITE_SIZE python3 $ROOT/bench.py $P --slot-order $so
  python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch_${so}_counter_collection.csv $OUT/pmc_write_${so}_counter_collection.csv $OUT/pmc_traffic_$so.json > /dev/null
  stats c3_${so}_one_in_flight python3 $ROOT/bench.py $B --in-flight 1 --slot-order $so
  stats c3_${so}_in_flight python3 $ROOT/bench.py $B --slot-order $so
done
stats c4_one_in_flight python3 $ROOT/bench.py --workload C4 --steps 100 --warmup 5 --cpu-seconds 0 --in-flight 1 --extras 0
stats c2_one_in_flight python3 $ROOT/bench.py --workload C2 --steps 300 --warmup 10 --cpu-seconds 0 --in-flight 1 --extras 0
stats c5_one_in_flight python3 $ROOT/bench.py --workload C5 --steps 100 --warmup 5 --cpu-seconds 0 --in-flight 1 --extras 0
stats lnn_unet python3 $ROOT/tools/bench_lnn.py --config kitti --steps 10 --warmup 3
stats lnn_scannet python3 $ROOT/tools/bench_lnn.py --config scannet --steps 6 --warmup 2
stats lnn_shapenet python3 $ROOT/tools/bench_lnn.py --config shapenet --steps 10 --warmup 3
cd $ROOT
for W in C3 C2 C4 C5; do
  python3 bench.py --workload $W --cpu-seconds 4 > $OUT/bench_$W.log 2>$OUT/bench_$W.err
  grep -h "^{\"metric\"" $OUT/bench_$W.log | tail -1 > $OUT/bench_${W}_line.json
  grep -h "^DETAILS " $OUT/bench_$W.log | tail -1 | cut -c9- > $OUT/bench_${W}_details.json
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.log 2>$OUT/bench_driver.err
grep -h "^{\"metric\"" $OUT/bench_driver.log | tail -1 > $OUT/bench_driver_line.json
grep -h "^DETAILS " $OUT/bench_driver.log | tail -1 | cut -c9- > $OUT/bench_driver_details.json
for t in kitti scannet shapenet; do python3 tools/bench_lnn.py --config $t --graph --steps 12 --warmup 4 2>&1 | tail -1; done > $OUT/lnn_graph_steps.txt
python3 tools/conv_time.py > $OUT/conv_time.txt 2>&1
python3 tools/conv_time.py --coarse 1 --shapes 64x64,128x128,128x64,96x96,192x192,256x256,256x128 > $OUT/conv_time_level2.txt 2>&1
bash tools/pmc_lds_c3.sh > $OUT/pmc_lds_c3.txt 2>&1
CONFIG=kitti bash tools/probes/r6_pmc_lds_unet.sh > $OUT/pmc_lds_kitti.txt 2>&1
rm -f $OUT/*.log
ls $OUT | head -60
for f in $OUT/bench_*_line.json; do echo "== $f"; cut -c1-400 $f; done
cat $OUT/lnn_graph_steps.txt
python3 - $OUT <<'PY'
import json, sys
for so in ("hash", "space"):
    t = json.load(open(f"{sys.argv[1]}/pmc_traffic_{so}.json"))
    tot = sum(v["traffic_bytes"] for v in t.values() if isinstance(v, dict))
    print(so, "sum of all launches of a step:", round(tot / 1e6, 1), "MB;", {k: round(v["traffic_bytes"] / 1e6, 1) for k, v in t.items() if isinstance(v, dict)})
PY
