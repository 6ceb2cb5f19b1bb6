#!/bin/bash
# A/B of library builds on ONE box, all chain kernels shown: VARIANTS="a b" (suffixes of liblatticenet_hip_<v>.so), product first, two rounds
for rep in 1 2; do
  for v in "" $VARIANTS; do
    lib=${v:+_$v}
    export LATTICE_NET_LIB=$(pwd)/lattice_net_amd/liblatticenet_hip$lib.so
    echo "== lib${lib:-_product} rep $rep: $(python tools/probes/r6_kernels.py hash 2>&1 | grep chain | cut -c17-) | $(python bench.py --steps 1500 --warmup 50 --cpu-seconds 0 --full-unet 0 --extras 0 2>/dev/null | tail -1 | cut -c90-110)"
  done
done
