#!/bin/bash
# A/B of the 16-row 64-channel convolution (line-shaped vs fragment-shaped gathers) on one box: tools/conv_time.py at both lattice sizes
for rep in 1 2; do
  for v in "" $VARIANTS; do
    lib=${v:+_$v}
    export LATTICE_NET_LIB=$(pwd)/lattice_net_amd/liblatticenet_hip$lib.so
    echo "== lib${lib:-_product} rep $rep"
    python tools/conv_time.py --shapes 64x64,64x128,64x32 2>&1 | grep "^V"
    python tools/conv_time.py --coarse 1 --shapes 64x64,64x128 2>&1 | grep "^V"
  done
done
