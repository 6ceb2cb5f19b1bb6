#!/bin/bash
# Round-2 quick check on the GPU box: new static-graph tests, then bench in both execution modes.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r2_check
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_static_graph.py -x -q -m gpu > $OUT/static_tests.log 2>&1; echo "static tests rc=$?"
tail -5 $OUT/static_tests.log
for mode in eager graph; do
  timeout 300 python bench.py --steps 50 --warmup 10 --cpu-seconds 0 --full-unet 0 --mode $mode > $OUT/bench_$mode.log 2>&1
  echo "bench $mode rc=$?"
  grep "^{" $OUT/bench_$mode.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'],'Mpts/s',d['ms_per_step'],'ms', d['config'].get('graph_vs_eager'), d['roofline'])
print(d['stages'])
" || tail -20 $OUT/bench_$mode.log
done
