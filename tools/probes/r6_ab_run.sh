#!/bin/bash
# round 6 A/B on one box: tests of the space-ordered build, then the bench line's per-kernel figures under both slot orders
mkdir -p gpurun_out/r6b
(timeout 1200 python -m pytest tests/test_gpu_space_order.py tests/test_gpu_slot_order.py tests/test_gpu_static_graph.py -x -q -m gpu 2>&1 | tail -30) > gpurun_out/r6b/tests.log
tail -15 gpurun_out/r6b/tests.log
for vw in ${VWS:-0}; do
  LATTICE_PLANE_VERTEX_WEIGHT=$vw timeout 400 python bench.py --steps 400 --warmup 20 --cpu-seconds 0 --full-unet 0 --ops-table 0 --slot-order space > gpurun_out/r6b/space_vw$vw.log 2>&1
  cp bench_details.json gpurun_out/r6b/space_vw$vw.json
  echo "== space vw=$vw"; python tools/probes/bench_summary.py gpurun_out/r6b/space_vw$vw.json; tail -3 gpurun_out/r6b/space_vw$vw.log | cut -c1-300
done
timeout 400 python bench.py --steps 400 --warmup 20 --cpu-seconds 0 --full-unet 0 --ops-table 0 --slot-order hash > gpurun_out/r6b/hash.log 2>&1
cp bench_details.json gpurun_out/r6b/hash.json
echo "== hash"; python tools/probes/bench_summary.py gpurun_out/r6b/hash.json
