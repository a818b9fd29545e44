set -e
mkdir -p gpurun_out
cd dragposer_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -fno-slp-vectorize -ffp-contract=on -DDP_PROFILE -o ../../gpurun_out/libdp_prof.so dp_host.cpp dp_kernel.hip dp_kernel4.hip
cd ../..
export DRAGPOSER_LIB=gpurun_out/libdp_prof.so
( python tools/profile_phases.py 4096 0
  DP_KERNEL=4x1 python tools/profile_phases.py 1024 6
  DP_KERNEL=4x1 python tools/profile_phases.py 4096 6
  DP_KERNEL=4x2 python tools/profile_phases.py 4096 6
  DP_KERNEL=4x2 python tools/profile_phases.py 8192 6 ) > gpurun_out/k4_phases.txt 2>&1
cat gpurun_out/k4_phases.txt
