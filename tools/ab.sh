#!/bin/bash
# usage (on the GPU box): LIBS="_scratch/libA.so _scratch/libB.so" [KERNEL=w4] [FRAMES=4096] tools/ab.sh
# Interleaved rounds of bench.py per library in one call: throughput (M frames/s) and kernel time per launch.
for rep in 1 2 3; do
for lib in $LIBS; do
  DP_KERNEL=${KERNEL:-} DRAGPOSER_LIB=$lib python3 bench.py --no-cpu-baseline --no-parity --frames ${FRAMES:-4096} --steps 50 | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(j['value']/1e6,3), 'M frames/s  kernel_ms', round(j['roofline']['kernel_ms'],5))"
done; done
