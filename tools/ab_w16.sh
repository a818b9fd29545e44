#!/bin/bash
# usage (on the GPU box): LIBS="_scratch/lib_a.so _scratch/lib_b.so" [SIZES="16384 65536"] [REPS=2] tools/ab_w16.sh  -- interleaved w16 timings per library
for rep in $(seq 1 ${REPS:-2}); do
for lib in $LIBS; do
  DRAGPOSER_LIB=$lib python3 tools/w16_sweep.py ${SIZES:-16384 65536} 2>/dev/null | grep "config 5" | sed "s#^#$lib #" | awk '{print $1, $8, "frames: w16", $22, "ms"}'
done; done
