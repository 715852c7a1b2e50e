#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "not variant_library" 2>&1 | tail -2
for w in n65536f n65536f cfg5; do python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().splitlines()[-1]);print(d['config']['workload'][:9],d['value'],round(d['roofline']['frac'],4),round(d['roofline']['kernel_ms'],3),d['parity_spot_check'])"; done
python tools/bench_other.py cfg5 2>&1 | grep -v amdgpu
