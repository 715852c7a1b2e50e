B="python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline"
P='import sys,json; d=json.loads(sys.stdin.read()); print("%.4g lines/s  %.3f ms  frac %.3f  ok=%s %s" % (d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["parity_spot_check"]["ok"], d["config"].get("options")))'
timeout -k 10 300 python -m pytest tests/test_gpu_large.py -q -k "not full_size" 2>&1 | tail -8
for r in 2 3 4; do timeout -k 10 200 $B --opt large_team=2 --opt large_ring=$r 2>/dev/null | python -c "$P"; done
for b in 16 12 64; do timeout -k 10 200 $B --opt large_team=2 --opt large_ring=3 --opt large_block=$b 2>/dev/null | python -c "$P"; done
