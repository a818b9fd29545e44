#!/bin/bash
# usage: tools/build_variant.sh NAME [extra hipcc flags...]
# Builds _scratch/libNAME.so from the CURRENT tree with extra flags (e.g. -DEXPERIMENT=1) applied to every source; for
# one-process A/B comparisons on the GPU box with tools/ab.sh (cdna_hip_programming.md 5.4 rule 24).  _scratch/ is git-ignored
# but travels with gpurun.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p _scratch/obj_$NAME
python3 - "$NAME" "$@" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
name, extra = sys.argv[1], sys.argv[2:]
objs, _ = g._compile_objects(True, extra_flags=extra, objdir=os.path.join(os.getcwd(), "_scratch", "obj_" + name), sources=g.HIP_SOURCES)
g._link(objs, os.path.join(os.getcwd(), "_scratch", "lib" + name + ".so"))
PY
ls -la _scratch/lib$NAME.so
