#!/bin/bash
# Regenerates the files under profiles/ (run on the GPU box through gpurun; results land in gpurun_out/prof_r06/).
# rocprofv3: program directly after `--`; counters in their own passes with --kernel-trace only.
set -e
export TMPDIR=/tmp
O=gpurun_out/prof_r06
rm -rf $O && mkdir -p $O
# 1. kernel-trace stats of the default bench command
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --steps 20 --no-cpu-baseline --traffic file > $O/bench_under_rocprof.json 2> $O/stats.log
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
echo "stats done"
# 2. PMC passes
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_F32" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  # (TCC budget: FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ: at most 8 per pass)
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i --output-format csv -- python3 bench.py --steps 10 --no-cpu-baseline --no-parity --traffic none > $O/pmc$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -5 $O/pmc$i.log; exit 1; }
  echo "pmc pass $i done"
done
python3 tools/pmc_summary.py $O/pmc_per_launch.json $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 $O/pmc6 $O/pmc7 > /dev/null
# 3. batch sweep
# (the library's own kernel choice: dp_w4 up to 8192 frames = two rounds, dp_w16 beyond; then the other kernel forced at the sizes around the threshold)
for spec in 256:auto 1024:auto 2048:auto 4096:auto 6144:auto 8192:auto 12288:auto 16384:auto 65536:auto 6144:w16 8192:w16 12288:w4 16384:w4 65536:w4; do
  fr=${spec%%:*}; kn=${spec##*:}
  python3 bench.py --frames $fr --kernel $kn --steps 20 --warmup 3 --no-cpu-baseline --no-parity --traffic none | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print($fr, '--kernel $kn:', '%.2f M frames/s' % (j['value']/1e6), 'frac %.4f' % r['frac'], 'kernel_ms %.4f' % r['kernel_ms'], r['kernel'])"
done > $O/batch_sweep.txt
echo "sweep done"
# 4. phase stamps (diagnostic build): the product's kernel, and the previous decomposition for comparison
( SPECS="4096:0:w4 1024:0:w4 4096:0:8" bash tools/phase_profile.sh ) > $O/phase_cycles.txt 2>&1
echo "phases done"
# 5. per-launch cost against the iteration count
python3 tools/time_iters.py 4096 > $O/time_vs_iters.txt 2>&1
# 6. the N > 1 path rehearsed on this one GPU: two ranks over gloo, both on device 0 (RCCL needs one GPU per rank), weak and
#    strong sizing; launched through torch.distributed.run exactly as the driver launches the real thing
for mode in "--frames 4096" "--total-frames 4096" "--total-frames 8192"; do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 \
      --backend gloo --all-ranks-on-device0 --no-cpu-baseline $mode 2>/dev/null | tail -1
done > $O/rehearsal_2ranks_gloo_one_gpu.jsonl
echo "rehearsal done"
# 7. the temporal predictor: times against PyTorch ops, and the kernel trace of its launches alone
python3 tools/time_temporal.py 2>&1 | grep "^window" > $O/temporal_predictor_times.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/tstats --output-format csv -- python3 tools/time_temporal.py --native-only > $O/tstats.log 2>&1
grep "dp_temporal_kernel\|\"Name\"" $(find $O/tstats -name "*kernel_stats.csv" | head -1) > $O/temporal_kernel_stats.csv
python3 tools/team_latency.py 1 2 4 8 16 32 64 128 2>&1 | grep "^window" > $O/team_latency.txt
if [ -f _scratch/lib_tstamps.so ]; then python3 tools/temporal_phases.py 1 0 --timeline 2>&1 | grep -v "amdgpu.ids\|Warning\|self.encoder" > $O/temporal_phases_team.txt; fi  # (tools/temporal_phases.sh on the build host first)
echo "temporal done"
# 8. the 16-frames-per-wave kernel: BASELINE config 5 through bench.py (the JSON line), kernel trace, batch sweep of both kernels, PMC passes
python3 bench.py --config s4 --frames 16384 --steps 20 --warmup 3 --no-cpu-baseline --traffic none > $O/bench_s4_16384.json 2> /dev/null
python3 bench.py --config s4 --frames 65536 --steps 10 --warmup 2 --no-cpu-baseline --traffic none > $O/bench_s4_65536.json 2> /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/s4stats --output-format csv -- python3 bench.py --config s4 --frames 16384 --steps 20 --no-cpu-baseline --traffic none > $O/bench_s4_under_rocprof.json 2> $O/s4stats.log
grep "dp_w16_kernel\|dp_w4_kernel\|\"Name\"" $(find $O/s4stats -name "*kernel_stats.csv" | head -1) > $O/bench_s4_kernel_stats.csv
python3 tools/w16_sweep.py 4096 5120 6144 8192 16384 32768 65536 131072 > $O/w16_sweep.txt 2>&1
bash tools/w16_pmc.sh 16384 $O/w16_pmc_16384 > $O/w16_pmc_16384.txt 2>&1
bash tools/w16_pmc.sh 65536 $O/w16_pmc_65536 > $O/w16_pmc_65536.txt 2>&1
python3 tools/w16_early_stop_timing.py 2>/dev/null > $O/w16_early_stop.txt
echo "w16 done"
# 9. sequences: the whole operator in lock-step, the cost of a step of a whole-sequence launch, the plug-in
python3 tools/throughput_sequences.py 6 2>&1 | grep "trackers" > $O/sequence_throughput.txt
python3 tools/throughput_sequences.py 3 2>&1 | grep "trackers" >> $O/sequence_throughput.txt
python3 tools/seq_timing.py 2>&1 | grep "S =\|steps in" > $O/sequence_step_cost.txt
python3 tools/plugin_latency.py > $O/plugin_latency.txt 2>&1 || true
echo "sequences done"
# 10. the default bench command as the driver runs it (cpu_baseline included), the smoke check and the GPU test log
python3 bench.py > $O/bench_default.json 2> /dev/null
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python3 -m pytest tests -q -m gpu -s 2>&1 | grep -v "amdgpu.ids\|Warning\|warnings.warn\|self.encoder" > $O/gpu_tests_log.txt || true
echo "bench + tests done"
# 11. round 5: the clock ramp (DVFS), the part-by-part ablation of dp_w4 at the steady clock (libraries built by tools/ablate_w4.sh on the build
#     host before this call), the dead-K-group A/B on the reference's three tracker configurations, the product's own closed-loop spread
python3 tools/clock_ramp.py 2>&1 | grep -v amdgpu.ids > $O/clock_ramp.txt || true
if ls _scratch/lib_ab_*.so > /dev/null 2>&1; then LIBS="$(ls _scratch/lib_ab_*.so | tr '\n' ' ')" REPS=3 bash tools/ab.sh > $O/ablation.txt 2>&1 || true; fi
if [ -f _scratch/lib_noskip.so ]; then REPS=3 python3 tools/ab_configs.py _scratch/lib_noskip.so dragposer_amd/lib/libdragposer_hip.so > $O/b2_skip_ab.txt 2>&1 || true; fi
python3 tools/clip_twins.py f1_clip4 f1_clip4_t 2>/dev/null | grep -v "^Mean\|^Time\|^Evaluate\|^Frames" > $O/clip_twins.txt || true
echo "round-5 probes done"
# 12. round 6: the headline launch's slope and intercept at the steady clock; L0's sub-phases pinned to the arrival of their results; the temporal
#     kernel's PAIR variant against the two-workgroups-per-CU one; the epilogue's share (library built by tools/build_variant_w4.sh ... -DW4_ABLATE_OUT)
python3 tools/launch_intercept.py dragposer_amd/lib/libdragposer_hip.so 2>&1 | grep -v amdgpu.ids > $O/launch_intercept.txt || true
if [ -f _scratch/lib_noout.so ]; then python3 tools/launch_intercept.py _scratch/lib_noout.so 2>&1 | grep -v amdgpu.ids >> $O/launch_intercept.txt || true; fi
( EXTRA_DEFS=-DW4_STAMP_L0 PHASE_L0=1 SPECS="4096:0:w4" bash tools/phase_profile.sh ) > $O/phase_cycles_L0.txt 2>&1 || true
python3 tools/ab_temporal_variants.py 42,44 2>&1 | grep "^window" > $O/temporal_pair_ab.txt || true
if [ -f _scratch/lib_nopk.so ]; then LIBS="_scratch/lib_nopk.so dragposer_amd/lib/libdragposer_hip.so" REPS=4 bash tools/ab.sh > $O/packed_adam_ab.txt 2>&1 || true; fi
echo "round-6 probes done"
ls $O
