#!/bin/bash
# in-flight throughput (bench.py --extras 0, 2000 steps) under experiment knobs given as "NAME=VALUE ..." per line on stdin
while read -r line; do
  [ -z "$line" ] && continue
  v=$(env $line timeout 300 python bench.py --steps 2000 --warmup 50 --extras 0 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$line -> $v"
done
