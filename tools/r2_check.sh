#!/bin/bash
# Round-2 quick check on the GPU box: bench in its execution modes (short runs, no CPU / U-Net legs).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r2_check
rm -rf $OUT; mkdir -p $OUT
for cfg in "graph2:--mode graph --in-flight 2" "graph1:--mode graph --in-flight 1" "eager:--mode eager" "graph4:--mode graph --in-flight 4"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --full-unet 0 $flags > $OUT/bench_$name.log 2>&1
  echo "bench $name rc=$?"
  grep "^{" $OUT/bench_$name.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
c=d['config']
print(' ', d['value'],'Mpts/s',d['ms_per_step'],'ms | graph_vs_eager', c.get('graph_vs_eager'), '| single', c.get('one_scan_in_flight'))
print('  roofline', d['roofline'])
print('  stages', d['stages']['us'])
" || tail -20 $OUT/bench_$name.log
done
