#!/bin/bash
# Diagnostic: dp_temporal_kernel with phase stamps (-DDPT_STAMPS) -> _scratch/lib_tstamps.so (other objects: the product build's);
# tools/temporal_phases.py then prints where one launch's cycles go.  Build on the build host, run the .py on the GPU box.
set -e
mkdir -p _scratch
B=dragposer_amd/csrc/_build
FLAGS=$(python3 -c "import __graft_entry__ as g; print(' '.join(f for f in g.HIPCC_FLAGS if f != '-shared'))")
hipcc $FLAGS -DDPT_STAMPS $EXTRA -c dragposer_amd/csrc/dp_temporal.hip -o _scratch/dp_temporal_stamps.o
hipcc --offload-arch=gfx950 -shared -fPIC -o _scratch/lib_tstamps.so $B/dp_host.o $B/dp_w16_host.o $B/dp_w4.o $B/dp_w16.o $B/dp_w16_2w.o $B/dp_w16_es.o $B/dp_w16_2w_es.o $B/dp_w16_long.o $B/dp_w16_2w_long.o $B/dp_w16_es_long.o $B/dp_w16_2w_es_long.o $B/dp_sequence.o _scratch/dp_temporal_stamps.o
echo _scratch/lib_tstamps.so
