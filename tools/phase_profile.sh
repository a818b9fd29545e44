# Diagnostic build with in-kernel phase stamps (-DDP_PROFILE) + tools/profile_phases.py for each "frames:hint:kernel" in
# $SPECS; $PWAVE selects the wave whose stamps are stored.  Flags and sources are the product build's (__graft_entry__.py).
set -e
mkdir -p gpurun_out
FLAGS=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIPCC_FLAGS))")
SRCS=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIP_SOURCES))")
( cd dragposer_amd/csrc && hipcc $FLAGS -DDP_PROFILE -DDP_PROFILE_WAVE=${PWAVE:-0} -o ../../gpurun_out/libdp_prof.so $SRCS )
export DRAGPOSER_LIB=gpurun_out/libdp_prof.so
for spec in ${SPECS:-"4096:0:8"}; do
  IFS=: read frames hint kern <<< "$spec"
  DP_KERNEL=$kern python3 tools/profile_phases.py $frames $hint
done > gpurun_out/phases.txt 2>&1
cat gpurun_out/phases.txt
