#!/bin/bash
run() { v=$1; w=$2; shift 2; SPEC_LIB_VARIANT=$v timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().splitlines()[-1]);print('$v $w $*', d['value'],round(d['roofline']['frac'],4),round(d['roofline']['kernel_ms'],3), d['parity_spot_check']['ok'])
except Exception as e: print('$v $w ERR', e)"; }
run "" n65536f
run tDP n65536f --opt large_wg=1024
run tDP n65536f --opt large_wg=1024 --opt large_ring=2
run tDP n65536f --opt large_wg=1024 --opt large_ring=1
run "" n65536f
run tDP n65536f --opt large_wg=1024
