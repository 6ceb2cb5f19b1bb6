#!/bin/bash
# Round-5 profile set, run on the GPU box (through gpurun): rocprofv3 kernel statistics of
#   bench.py (default: four scans in flight; and --in-flight 1), the per-operator table (bench.py --workload ops), the whole-network
#   step (tools/bench_lnn.py), the C5 line;
# the two HBM-traffic PMC passes of the C3 step, and the matrix-core PMC passes (MFMA busy cycles + MOPS by type) over the C3 step,
# the per-operator table and C5.  Every --pmc pass is its own run with --kernel-trace only (MI355X_MICROARCH.md).
# Outputs land in gpurun_out/prof_r5/ ; tools/gpu_profile_r5_collect.py copies the summaries into profiles/r5_*.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r5
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="--steps 300 --warmup 10 --cpu-seconds 0 --full-unet 0 --extras 0"
stats() {  # name, command...
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- "$@" > $OUT/$name.log 2>&1
  local s=$(find $OUT/$name -name "*kernel_stats.csv" | head -1); cp "$s" $OUT/${name}_kernel_stats.csv 2>/dev/null
  grep -h "^{\"metric\"" $OUT/$name.log | tail -1 > $OUT/${name}_bench_line.json 2>/dev/null
  grep -h "^DETAILS " $OUT/$name.log | tail -1 | cut -c9- > $OUT/${name}_bench_details.json 2>/dev/null
  rm -rf $OUT/$name
}
pmc() {  # name, counters, command...
  local name=$1; local ctr=$2; shift; shift
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$name -o $name -- "$@" > $OUT/$name.log 2>&1
  local s=$(find $OUT/$name -name "*counter_collection.csv" | head -1); cp "$s" $OUT/${name}_counter_collection.csv 2>/dev/null
  rm -rf $OUT/$name
}
stats c3_in_flight python3 $ROOT/bench.py $B
stats c3_one_in_flight python3 $ROOT/bench.py $B --in-flight 1
stats ops python3 $ROOT/bench.py --workload ops
stats lnn_unet python3 $ROOT/tools/bench_lnn.py --config kitti --steps 10 --warmup 3
stats c5 python3 $ROOT/bench.py --workload C5 --steps 100 --warmup 5 --cpu-seconds 0 --full-unet 0 --extras 0 --in-flight 1
P="--steps 8 --warmup 2 --cpu-seconds 0 --full-unet 0 --extras 0 --in-flight 1"
pmc pmc_fetch FETCH_SIZE python3 $ROOT/bench.py $P
pmc pmc_write WRITE_SIZE python3 $ROOT/bench.py $P
pmc mfma_busy_c3 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" python3 $ROOT/bench.py $P
pmc mfma_mops_c3 "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32" python3 $ROOT/bench.py $P
pmc mfma_busy_ops "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" python3 $ROOT/tools/ops_roofline.py --reps 6
pmc mfma_mops_ops "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32" python3 $ROOT/tools/ops_roofline.py --reps 6
P5="--workload C5 --steps 8 --warmup 2 --cpu-seconds 0 --full-unet 0 --extras 0 --in-flight 1"
pmc mfma_busy_c5 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" python3 $ROOT/bench.py $P5
pmc mfma_mops_c5 "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32" python3 $ROOT/bench.py $P5
pmc lds_ops "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" python3 $ROOT/tools/ops_roofline.py --reps 6
pmc lds_c5 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" python3 $ROOT/bench.py $P5
pmc clock_ops "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" python3 $ROOT/tools/ops_roofline.py --reps 6
cd $ROOT
# the bench lines of every workload, unprofiled (the compact last line + the DETAILS line)
for W in C3 C2 C4 C5; do
  python3 bench.py --workload $W --cpu-seconds 4 > $OUT/bench_$W.log 2>$OUT/bench_$W.err
  grep -h "^{\"metric\"" $OUT/bench_$W.log | tail -1 > $OUT/bench_${W}_line.json
  grep -h "^DETAILS " $OUT/bench_$W.log | tail -1 | cut -c9- > $OUT/bench_${W}_details.json
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.log 2>$OUT/bench_driver.err
grep -h "^{\"metric\"" $OUT/bench_driver.log | tail -1 > $OUT/bench_driver_line.json
grep -h "^DETAILS " $OUT/bench_driver.log | tail -1 | cut -c9- > $OUT/bench_driver_details.json
python3 tools/conv_time.py > $OUT/conv_time.txt 2>&1
python3 tools/probes/gf_time.py > $OUT/gf_time.txt 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_fetch_counter_collection.csv $OUT/pmc_write_counter_collection.csv $OUT/pmc_traffic.json > /dev/null
python3 tools/gpu_profile_r5_collect.py $OUT --summary
