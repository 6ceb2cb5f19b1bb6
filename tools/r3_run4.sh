set -u
mkdir -p gpurun_out/r3d
LATTICE_NET_LIB=$PWD/lattice_net_amd/liblatticenet_hip_stamps.so timeout 300 python tools/kernel_timeline.py > gpurun_out/r3d/timeline.txt 2>&1; tail -14 gpurun_out/r3d/timeline.txt
LN_DEBUG_MASK=8 LATTICE_NET_LIB=$PWD/lattice_net_amd/liblatticenet_hip_stamps.so timeout 300 python tools/kernel_timeline.py > gpurun_out/r3d/timeline_noatomics.txt 2>&1; tail -14 gpurun_out/r3d/timeline_noatomics.txt
