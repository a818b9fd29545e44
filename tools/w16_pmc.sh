#!/bin/bash
# PMC passes over launches of the 16-frames-per-wave kernel (on the GPU box): tools/w16_pmc.sh FRAMES OUTDIR
set -e
export TMPDIR=/tmp
B=${1:-65536}; O=${2:-gpurun_out/w16_pmc_$B}
rm -rf $O && mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i --output-format csv -- python3 tools/w16_run.py w16 $B 4 > $O/pmc$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -5 $O/pmc$i.log; }
  echo "pass $i done"
done
python3 - $O <<'PY'
import sys, glob, csv, collections
O = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dp_w16_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:32s} launches {len(v)}  mean {sum(v)/len(v):16.1f}")
PY
