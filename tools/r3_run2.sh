set -u
mkdir -p gpurun_out/r3b
LATTICE_NET_LIB=$PWD/lattice_net_amd/liblatticenet_hip_stamps.so timeout 300 python tools/kernel_timeline.py > gpurun_out/r3b/timeline.txt 2>&1; head -30 gpurun_out/r3b/timeline.txt
timeout 600 python tools/chain_inflight.py --in-flight 1,2,3,4,6 --prefetch 0 > gpurun_out/r3b/chain_p0.txt 2>&1; tail -6 gpurun_out/r3b/chain_p0.txt
timeout 600 python tools/chain_inflight.py --in-flight 1,3 --prefetch 1 > gpurun_out/r3b/chain_p1.txt 2>&1; tail -3 gpurun_out/r3b/chain_p1.txt
