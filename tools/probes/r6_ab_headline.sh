#!/bin/bash
# headline only (four scans in flight), several library builds x environment settings on ONE box: VARIANTS="a b", ENVS="X=1 Y=2" (each tried alone)
for rep in 1 2; do
  for v in "" $VARIANTS; do
    lib=${v:+_$v}
    export LATTICE_NET_LIB=$(pwd)/lattice_net_amd/liblatticenet_hip$lib.so
    for e in "_" $ENVS; do
      if [ "$e" = "_" ]; then r=$(python bench.py --steps 1500 --warmup 50 --cpu-seconds 0 --full-unet 0 --extras 0 2>/dev/null | tail -1 | cut -c90-106)
      else r=$(env $e python bench.py --steps 1500 --warmup 50 --cpu-seconds 0 --full-unet 0 --extras 0 2>/dev/null | tail -1 | cut -c90-106); fi
      echo "== lib${lib:-_product} $e rep $rep: $r"
    done
  done
done
