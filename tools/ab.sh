#!/bin/bash
# usage (on the GPU box): LIBS="_scratch/libA.so _scratch/libB.so" [FRAMES=4096] [REPS=3] tools/ab.sh
# Interleaved rounds of bench.py per library in one call: throughput (M frames/s) and kernel time per launch; the last
# lines give each library's median kernel time (boxes differ by +-2 %, so compare within one call only).
for rep in $(seq 1 ${REPS:-3}); do
for lib in $LIBS; do
  python3 bench.py --lib $lib --no-cpu-baseline --no-parity --traffic none --frames ${FRAMES:-4096} ${EXTRA} --steps 50 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(j['value']/1e6,3), 'M frames/s  kernel_ms', round(j['roofline']['kernel_ms'],5))"
done; done | tee /tmp/ab_$$.txt
python3 - /tmp/ab_$$.txt <<'PY'
import sys, statistics, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.split()
    d[p[0]].append(float(p[-1]))
for k, v in d.items():
    print(f"median {k}: {statistics.median(v):.5f} ms  (min {min(v):.5f})")
PY
