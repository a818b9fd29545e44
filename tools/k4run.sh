set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_kernel4.py -x -q -m gpu > gpurun_out/k4_tests.log 2>&1 || { tail -30 gpurun_out/k4_tests.log; exit 1; }
tail -3 gpurun_out/k4_tests.log
for fr in 1024 2048 4096 8192 16384 65536; do
  for k in 8 4x1 4x2; do
    echo "== frames $fr kernel $k" >> gpurun_out/k4_bench.log
    DP_KERNEL=$k timeout -k 10 120 python bench.py --frames $fr --steps 20 --warmup 3 --no-cpu-baseline --no-parity >> gpurun_out/k4_bench.log 2>&1
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/k4_bench.log"):
    if l.startswith("=="): print(l.strip(), end="  ")
    elif l.startswith("{"):
        j=json.loads(l); print(f"{j['value']/1e6:.2f} Mframes/s  kern_ms {j['roofline']['kernel_ms']:.4f} frac {j['roofline']['frac']:.3f} {j['roofline']['kernel']}")
PY
