#!/bin/bash
# Instruction-issue and matrix-pipe counters of the convolution kernels (run on the GPU box through gpurun):
#   tools/pmc_conv.sh <tag> [conv_time.py arguments]      -> gpurun_out/pmc_conv_<tag>.json (+ the raw csv files)
# Every --pmc pass is its own rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md); python3 is the profiled program.
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_conv_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pass() {
  local name=$1; local ctr=$2; shift; shift
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$name -o $name -- python3 $ROOT/tools/conv_time.py --reps 3 "$@" > $OUT/$name.log 2>&1
  local s=$(find $OUT/$name -name "*counter_collection.csv" | head -1); cp "$s" $OUT/${name}.csv 2>/dev/null
  rm -rf $OUT/$name
}
pass busy "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "$@"
pass insts "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@"
pass mops "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "$@"
cd $ROOT
python3 tools/pmc_conv_collect.py $OUT gpurun_out/pmc_conv_$TAG.json
