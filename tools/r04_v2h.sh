#!/bin/bash
# round-4 development run: the 32768-point single-workgroup kernel -- tests, then timings (product + PF variants)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_v2h.py -m gpu -x -q > gpurun_out/r04_v2h_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r04_v2h_tests.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 200 python tools/bench_v2h.py base > gpurun_out/r04_v2h_bench.log 2>&1 || exit $?
for v in v2hpf0 v2hpf8 v2hpf20; do
  echo "variant $v" >> gpurun_out/r04_v2h_bench.log
  SPEC_LIB_VARIANT=$v timeout -k 10 100 python tools/bench_v2h.py pf >> gpurun_out/r04_v2h_bench.log 2>&1 || exit $?
done
timeout -k 10 100 python tools/bench_v2h.py lpw >> gpurun_out/r04_v2h_bench.log 2>&1
cat gpurun_out/r04_v2h_bench.log
