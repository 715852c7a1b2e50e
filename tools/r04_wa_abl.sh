#!/bin/bash
# ablation timings of the wave-autonomous column side (results wrong by construction)
for lib in "$@"; do
  echo "== library: $lib"
  SPEC_LIB_VARIANT=$lib timeout -k 10 300 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --opt large_team=2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 %.3f ms' % d['roofline']['kernel_ms'])" || echo failed
done
