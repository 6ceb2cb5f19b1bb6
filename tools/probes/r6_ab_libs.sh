#!/bin/bash
# A/B of several builds of the library on ONE box: VARIANTS="a b c" (suffixes of liblatticenet_hip_<v>.so), product first, two rounds
for rep in 1 2; do
  for v in "" $VARIANTS; do
    lib=${v:+_$v}
    export LATTICE_NET_LIB=$(pwd)/lattice_net_amd/liblatticenet_hip$lib.so
    echo "== lib${lib:-_product} rep $rep: $(python tools/probes/r6_kernels.py hash 2>&1 | grep chain | sed -e 's/.*k_conv_mfma/fwd/' -e 's/k_slice_forward.*k_conv_backward_fused/bwd/' -e 's/k_reduce_slabs.*sum/sum/') | $(python bench.py --steps 1500 --warmup 50 --cpu-seconds 0 --full-unet 0 --extras 0 2>/dev/null | tail -1 | cut -c90-110)"
  done
done
