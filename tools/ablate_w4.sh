#!/bin/bash
# Diagnostic (on the build host, then `LIBS="..." tools/ab.sh` on the GPU box): dp_w4 with ONE part of the iteration left out per
# library, to time the rest without stamps (s_memtime stamps attribute a phase's vector tail to the next phase: DESIGN.md 5.3).
# The ablated builds compute nonsense; only their launch time is read.
set -e
for v in ADAM T G J QT; do SKIP_HAZARD_CHECK=1 bash tools/build_variant_w4.sh ab_$v "-mllvm -amdgpu-sched-strategy=max-ilp -DW4_ABLATE_$v" > /dev/null; done
SKIP_HAZARD_CHECK=1 bash tools/build_variant_w4.sh ab_TG "-mllvm -amdgpu-sched-strategy=max-ilp -DW4_ABLATE_T -DW4_ABLATE_G" > /dev/null
SKIP_HAZARD_CHECK=1 bash tools/build_variant_w4.sh ab_JTG "-mllvm -amdgpu-sched-strategy=max-ilp -DW4_ABLATE_J -DW4_ABLATE_T -DW4_ABLATE_G" > /dev/null
bash tools/build_variant_w4.sh ab_none "-mllvm -amdgpu-sched-strategy=max-ilp" > /dev/null
ls _scratch/lib_ab_*.so
