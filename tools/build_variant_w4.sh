#!/bin/bash
# usage: tools/build_variant_w4.sh NAME "hipcc flags for dp_w4.hip (replacing the scheduling strategy)" -> _scratch/lib_NAME.so (other objects: the product build's)
set -e
mkdir -p _scratch
B=dragposer_amd/csrc/_build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize -ffp-contract=on"
hipcc $FLAGS $2 -c dragposer_amd/csrc/dp_w4.hip -o _scratch/dp_w4_$1.o
# the hand-padded MFMA groups rely on the wait states hipcc puts between asm statements: hold THIS variant's ISA to the requirement
# (diagnostic builds that leave parts out -- W4_ABLATE_* -- compute nonsense anyway: SKIP_HAZARD_CHECK=1)
if [ -z "$SKIP_HAZARD_CHECK" ]; then python3 tools/check_mfma_hazards.py --flags "$FLAGS $2" >&2; fi
hipcc --offload-arch=gfx950 -shared -fPIC -o _scratch/lib_$1.so $B/dp_host.o $B/dp_w16_host.o _scratch/dp_w4_$1.o $B/dp_w16.o $B/dp_w16_2w.o $B/dp_w16_es.o $B/dp_w16_2w_es.o $B/dp_w16_long.o $B/dp_w16_2w_long.o $B/dp_w16_es_long.o $B/dp_w16_2w_es_long.o $B/dp_sequence.o $B/dp_temporal.o
echo _scratch/lib_$1.so
