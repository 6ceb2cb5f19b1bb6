set -u
mkdir -p gpurun_out/r3c
timeout 600 python -m pytest tests/test_gpu_slot_order.py -x -q -m gpu > gpurun_out/r3c/slot.txt 2>&1; tail -15 gpurun_out/r3c/slot.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_build_paths.py tests/test_gpu_static_graph.py tests/test_gpu_surface.py -x -q -m gpu > gpurun_out/r3c/parity.txt 2>&1; tail -5 gpurun_out/r3c/parity.txt
timeout 300 python tools/profile_kernels.py --steps 20 > gpurun_out/r3c/prof.txt 2>&1; cat gpurun_out/r3c/prof.txt
timeout 600 python tools/chain_inflight.py --in-flight 1,2,3,4 --prefetch 0 > gpurun_out/r3c/chain_p0.txt 2>&1; tail -4 gpurun_out/r3c/chain_p0.txt
LN_DEBUG_MASK=4 timeout 600 python tools/chain_inflight.py --in-flight 1,3 --prefetch 0 > gpurun_out/r3c/chain_p0_nofuse.txt 2>&1; tail -2 gpurun_out/r3c/chain_p0_nofuse.txt
