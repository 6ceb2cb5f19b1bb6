#!/bin/bash
# C4 (129 k vertices) under forced sub-tile counts of the fused backward: one scan at a time and four in flight
for t in 0 1 2 3; do
  for f in 1 4; do
    echo "== LN_BWD_T=$t in-flight $f: $(LN_BWD_T=$t python bench.py --workload C4 --steps 400 --warmup 20 --cpu-seconds 0 --extras 0 --in-flight $f 2>/dev/null | tail -1 | cut -c90-130)"
  done
done
