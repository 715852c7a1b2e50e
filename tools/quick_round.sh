mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s3_pytest.txt 2>&1; tail -3 gpurun_out/s3_pytest.txt
for w in cfg2 cfg3 cfg4 n16384 n32768f n65536f; do
  timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --no-live-traffic > gpurun_out/s3_bench_$w.json 2> gpurun_out/s3_bench_$w.err || echo "bench $w failed"
  python - <<PY
import json
d=json.load(open("gpurun_out/s3_bench_$w.json"))
print("$w", "value %.4g"%d["value"], "kernel_ms %.3f"%d["roofline"]["kernel_ms"], "frac %.3f"%d["roofline"]["frac"], "sclk", d["roofline"]["clocks"]["sclk_mhz"]["median"], "spot", d["parity_spot_check"]["ok"])
PY
done
timeout -k 10 200 python tools/bench_other.py sizes > gpurun_out/s3_other.txt 2>&1; cat gpurun_out/s3_other.txt
