#!/bin/bash
# throughput A/B: bench.py (no profiler) for configurations "name:ENV=.. ENV=..", several in-flight counts
set -u
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for B in ${TP_INFLIGHT:-1 2 3}; do
    ( for kv in $envs; do export "$kv"; done
      timeout 120 python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --full-unet 0 --extra-kernels "" --in-flight $B 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$name B=$B', d['value'], 'Mpts/s', round(d['ms_per_step']*1e3,1), 'us/scan')" )
  done
done
