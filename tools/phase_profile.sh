set -e
mkdir -p gpurun_out
cd dragposer_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -fno-slp-vectorize -ffp-contract=on -DDP_PROFILE -DDP_PROFILE_WAVE=${PWAVE:-0} -o ../../gpurun_out/libdp_prof.so dp_host.cpp dp_kernel.hip dp_kernel4.hip
cd ../..
export DRAGPOSER_LIB=gpurun_out/libdp_prof.so
for spec in ${SPECS:-"4096:0:8"}; do
  IFS=: read frames hint kern <<< "$spec"
  DP_KERNEL=$kern python tools/profile_phases.py $frames $hint
done > gpurun_out/phases.txt 2>&1
cat gpurun_out/phases.txt
