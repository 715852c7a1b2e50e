#!/bin/bash
# round-3 experiment 3: which side of the team kernel should ask late (variants tB tC tD), unpaired 256-thread geometry (tE)
run() { v=$1; w=$2; shift 2; SPEC_LIB_VARIANT=$v timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().splitlines()[-1]);print('$v $w $*', d['value'],round(d['roofline']['frac'],4),round(d['roofline']['kernel_ms'],3))
except Exception as e: print('$v $w ERR', e)"; }
for rep in 1 2; do
for w in cfg5 n65536f; do
  run "" $w
  run tB $w
  run tC $w
  run tD $w
done
done
run tE cfg5 --opt large_wg=256
run tE n65536f --opt large_wg=256
run tE n65536f --opt large_wg=256 --opt large_ring=3
run "" n65536f --opt large_ring=3
run tB n65536f --opt large_ring=3
