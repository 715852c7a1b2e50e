#!/bin/bash
mkdir -p gpurun_out/exp2
timeout -k 10 600 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "not variant_library" > gpurun_out/exp2/tests.log 2>&1; tail -3 gpurun_out/exp2/tests.log
for w in cfg5 n65536f; do python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().splitlines()[-1]);print(d['config']['workload'][:8],d['value'],d['roofline']['frac'],d['roofline']['kernel_ms'],d['ms_per_step'])"; done
bash tools/r03_prof.sh
