#!/bin/bash
# Diagnostic: dp_w16 with ONE part of the iteration left out per library (-DW16_ABLATE_*: the ablated builds compute nonsense, only their
# time is read -- the method of tools/ablate_w4.sh).  Builds _scratch/lib_a16_{none,BPERM,SPLIT,MM5,KIN,T}.so; time them on the GPU box with
#   LIBS="_scratch/lib_a16_none.so ..." SIZES="16384 65536" REPS=3 tools/ab_w16.sh
S1="-mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -amdgpu-mfma-vgpr-form"
ES=0 tools/build_variant_w16.sh a16_none "$S1" ""
for part in BPERM SPLIT MM5 KIN T; do
  ES=0 tools/build_variant_w16.sh a16_$part "$S1 -DW16_ABLATE_$part" "-DW16_ABLATE_$part"
done
