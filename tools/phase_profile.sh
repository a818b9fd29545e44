# Diagnostic build with in-kernel phase stamps (-DDP_PROFILE) + tools/profile_phases.py for each "frames:hint:kernel" in
# $SPECS (kernel: w4 = the product's, 8 = the test-only 8-wave kernel: dp_host.cpp -DDP_REF8_BUILD + dp_kernel.hip); $PWAVE selects
# the wave whose stamps are stored.  Flags and sources are the product build's (__graft_entry__.py).
set -e
mkdir -p gpurun_out
FLAGS=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIPCC_FLAGS))")
SRCS=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIP_SOURCES))")
( cd dragposer_amd/csrc && hipcc $FLAGS -DDP_PROFILE ${EXTRA_DEFS} -DDP_PROFILE_WAVE=${PWAVE:-0} -o ../../gpurun_out/libdp_prof.so $SRCS )
( cd dragposer_amd/csrc && hipcc $FLAGS -DDP_PROFILE -DDP_PROFILE_WAVE=${PWAVE:-0} -DDP_REF8_BUILD -o ../../gpurun_out/libdp_prof8.so $SRCS dp_kernel.hip )
for spec in ${SPECS:-"4096:0:w4"}; do
  IFS=: read frames hint kern <<< "$spec"
  if [ "$kern" = "8" ]; then lib=gpurun_out/libdp_prof8.so; else lib=gpurun_out/libdp_prof.so; fi
  DRAGPOSER_LIB=$lib PHASE_KERNEL=$kern python3 tools/profile_phases.py $frames $hint
done > gpurun_out/phases.txt 2>&1
cat gpurun_out/phases.txt
