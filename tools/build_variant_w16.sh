#!/bin/bash
# usage: tools/build_variant_w16.sh NAME "flags for the one-wave-per-SIMD units" ["flags for the two-waves units"] -> _scratch/lib_NAME.so
# (the flags REPLACE the scheduling strategy of the product build for dp_w16.hip + dp_w16_es.hip / dp_w16_2w.hip + dp_w16_2w_es.hip -- the
#  early-stop instantiations are rebuilt with the same flags as their fixed-count twins, so an A/B covers all four kernels; ES=0 keeps the
#  product build's early-stop objects; the other objects are the product build's)
set -e
mkdir -p _scratch
B=dragposer_amd/csrc/_build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize -ffp-contract=on"
hipcc $FLAGS $2 -c dragposer_amd/csrc/dp_w16.hip -o _scratch/dp_w16_$1.o
hipcc $FLAGS $3 -c dragposer_amd/csrc/dp_w16_2w.hip -o _scratch/dp_w16_2w_$1.o
ES1=$B/dp_w16_es.o; ES2=$B/dp_w16_2w_es.o
if [ "${ES:-1}" != "0" ]; then
  hipcc $FLAGS $2 -c dragposer_amd/csrc/dp_w16_es.hip -o _scratch/dp_w16_es_$1.o
  hipcc $FLAGS $3 -c dragposer_amd/csrc/dp_w16_2w_es.hip -o _scratch/dp_w16_2w_es_$1.o
  ES1=_scratch/dp_w16_es_$1.o; ES2=_scratch/dp_w16_2w_es_$1.o
fi
hipcc --offload-arch=gfx950 -shared -fPIC -o _scratch/lib_$1.so $B/dp_host.o $B/dp_w16_host.o $B/dp_w4.o _scratch/dp_w16_$1.o _scratch/dp_w16_2w_$1.o $ES1 $ES2 $B/dp_w16_long.o $B/dp_w16_2w_long.o $B/dp_w16_es_long.o $B/dp_w16_2w_es_long.o $B/dp_sequence.o $B/dp_temporal.o
echo _scratch/lib_$1.so
