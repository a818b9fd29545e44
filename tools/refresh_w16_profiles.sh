set -e
export TMPDIR=/tmp
O=gpurun_out/prof_r03b; rm -rf $O; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python3 bench.py --steps 20 --warmup 3 > $O/bench_default.json 2>/dev/null; python3 -c "import json; j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default bench', round(j['value']/1e6,2), j['roofline']['frac'], j['cpu_baseline']['value'])"
python3 bench.py --config s4 --frames 16384 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_s4_16384.json 2> /dev/null
python3 bench.py --config s4 --frames 65536 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_s4_65536.json 2> /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/s4stats --output-format csv -- python3 bench.py --config s4 --frames 16384 --steps 20 --no-cpu-baseline > $O/bench_s4_under_rocprof.json 2> $O/s4stats.log
grep "dp_w16_kernel\|dp_w4_kernel\|\"Name\"" $(find $O/s4stats -name "*kernel_stats.csv" | head -1) > $O/bench_s4_kernel_stats.csv
python3 tools/w16_sweep.py 4096 8192 16384 32768 65536 131072 > $O/w16_sweep.txt 2>&1
bash tools/w16_pmc.sh 16384 $O/w16_pmc_16384 > $O/w16_pmc_16384.txt 2>&1
bash tools/w16_pmc.sh 65536 $O/w16_pmc_65536 > $O/w16_pmc_65536.txt 2>&1
echo done; ls $O
