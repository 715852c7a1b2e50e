#!/bin/bash
# round-3 experiment 1: baselines on this box + team-kernel knobs for fp32 lines
set -u
mkdir -p gpurun_out/exp1
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline"
run() { name=$1; shift; timeout -k 10 240 $B "$@" > gpurun_out/exp1/$name.json 2> gpurun_out/exp1/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/exp1/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))
except Exception as e: print('ERR', e)
PY
)"; }
run cfg5 --workload cfg5
run n65536f --workload n65536f
run n65536f_dense --workload n65536f --opt large_wg=1024
run n65536f_wg256 --workload n65536f --opt large_wg=256
run n65536f_ring3 --workload n65536f --opt large_ring=3
run n65536f_two --workload n65536f --opt large_team=0
run cfg5_dense --workload cfg5 --opt large_wg=1024
run cfg4 --workload cfg4
run cfg3 --workload cfg3
run cfg2 --workload cfg2
run n16384 --workload n16384
