#!/bin/bash
# Regenerates the files under profiles/ (run on the GPU box through gpurun; results land in gpurun_out/prof_r01/).
# rocprofv3: program directly after `--`; counters in their own passes with --kernel-trace only.
set -e
export TMPDIR=/tmp
O=gpurun_out/prof_r01
rm -rf $O && mkdir -p $O
# 1. kernel-trace stats of the default bench command
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --steps 20 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
echo "stats done"
# 2. PMC passes
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_F32" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  # (TCC budget: FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ: at most 8 per pass)
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i --output-format csv -- python3 bench.py --steps 10 --no-cpu-baseline --no-parity > $O/pmc$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -5 $O/pmc$i.log; exit 1; }
  echo "pmc pass $i done"
done
python3 tools/pmc_summary.py $O/pmc_per_launch.json $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 $O/pmc6 $O/pmc7 > /dev/null
# 3. batch sweep (library's own dispatch, hint 6)
for fr in 1024 2048 4096 8192 16384 65536; do
  python3 bench.py --frames $fr --steps 20 --warmup 3 --no-cpu-baseline --no-parity | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print($fr, '%.2f M frames/s' % (j['value']/1e6), 'frac %.4f' % r['frac'], 'kernel_ms %.4f' % r['kernel_ms'], r['kernel'])"
done > $O/batch_sweep.txt
echo "sweep done"
# 4. phase stamps (diagnostic build)
( PWAVE=0 SPECS="4096:0:8 1024:6:4x1 4096:6:4x1 4096:6:4x2 8192:6:4x2" bash tools/phase_profile.sh; echo "---- stamps of wave 4 (second-dispatched half)"; PWAVE=4 SPECS="4096:0:8" bash tools/phase_profile.sh ) > $O/phase_cycles_variants.txt 2>&1
echo "phases done"
ls $O
