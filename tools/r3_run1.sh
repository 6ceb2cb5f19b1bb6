set -u
mkdir -p gpurun_out/r3a
timeout 600 python -m pytest tests/test_gpu_slot_order.py -x -q -m gpu > gpurun_out/r3a/slot.txt 2>&1; tail -15 gpurun_out/r3a/slot.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_build_paths.py tests/test_gpu_static_graph.py -x -q -m gpu > gpurun_out/r3a/parity.txt 2>&1; tail -15 gpurun_out/r3a/parity.txt
timeout 300 python tools/profile_kernels.py --steps 20 > gpurun_out/r3a/prof.txt 2>&1; cat gpurun_out/r3a/prof.txt
