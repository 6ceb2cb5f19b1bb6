#!/bin/bash
# A/B of two builds of the library on ONE box: chain kernel times and the headline, alternating (VARIANT = suffix of liblatticenet_hip_<VARIANT>.so)
V=${VARIANT:-frag}
for rep in 1 2; do
  for lib in "" "_$V"; do
    export LATTICE_NET_LIB=$(pwd)/lattice_net_amd/liblatticenet_hip$lib.so
    echo "== lib${lib:-_product} rep $rep"
    python tools/probes/r6_kernels.py hash 2>&1 | grep "chain"
    python bench.py --steps 1500 --warmup 50 --cpu-seconds 0 --full-unet 0 --extras 0 2>/dev/null | tail -1 | cut -c90-140
  done
done
