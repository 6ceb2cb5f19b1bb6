#!/bin/bash
# secondary workloads under several library builds on ONE box: VARIANTS="a b"; in flight (the bench line) and one scan at a time (latency)
for w in C4 C5 C2; do
  for v in "" $VARIANTS; do
    lib=${v:+_$v}
    export LATTICE_NET_LIB=$(pwd)/lattice_net_amd/liblatticenet_hip$lib.so
    a=$(python bench.py --workload $w --steps 600 --warmup 20 --cpu-seconds 0 --extras 0 2>/dev/null | tail -1 | cut -c90-106)
    b=$(python bench.py --workload $w --steps 300 --warmup 20 --cpu-seconds 0 --extras 0 --in-flight 1 2>/dev/null | tail -1 | cut -c90-106)
    echo "== $w lib${lib:-_product}: in flight $a | one at a time $b"
  done
done
for v in "" $VARIANTS; do
  lib=${v:+_$v}
  export LATTICE_NET_LIB=$(pwd)/lattice_net_amd/liblatticenet_hip$lib.so
  for t in kitti scannet; do echo "== lib${lib:-_product} $(python tools/bench_lnn.py --config $t --graph --steps 12 --warmup 4 2>&1 | tail -1 | cut -c1-80)"; done
done
