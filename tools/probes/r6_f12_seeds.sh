#!/bin/bash
# which parameter seed of the F12 fixture stays on the no-flip side in deterministic mode (and in default mode)?
cp tests/golden/F12_reference_lnn_kitti.npz /tmp/F12_orig.npz
for f in tests/golden/tmp_f12/F12_*.npz /tmp/F12_orig.npz; do
  cp $f tests/golden/F12_reference_lnn_kitti.npz
  echo "== $f"
  RUNS=2 python tools/probes/r6_determinism.py 2>&1 | grep "run "
done
cp /tmp/F12_orig.npz tests/golden/F12_reference_lnn_kitti.npz
