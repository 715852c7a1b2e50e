#!/bin/bash
# round-4 experiment: wave-autonomous team kernel (variant library tWA) -- parity first, then cfg5 timings beside the product
set -o pipefail
mkdir -p gpurun_out
V=${1:-tWA}
SPEC_LIB_VARIANT=$V timeout -k 10 600 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "(team_kernel_matches_oracle or ring_depths or two_launch_paths_agree or repeated) and not wg256" > gpurun_out/r04_wa_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r04_wa_tests.log
[ $rc -ne 0 ] && exit $rc
for lib in "" $V; do
  echo "== library: ${lib:-product}"
  SPEC_LIB_VARIANT=$lib timeout -k 10 300 python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 %.3f ms  frac %.3f  parity %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['parity_spot_check']['ok']))" || exit 1
done
