#!/bin/bash
# round-4: multi-context / multi-rank self-checks on one GPU (three contexts on device 0; two gloo ranks on device 0)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "multi" > gpurun_out/r04_multi_tests.log 2>&1
rc=$?; tail -3 gpurun_out/r04_multi_tests.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python tools/bench_multi.py --contexts 3 --log2-samples 27 > gpurun_out/r04_bench_multi_3ctx.json 2> gpurun_out/r04_bench_multi.err || { tail -5 gpurun_out/r04_bench_multi.err; exit 1; }
cat gpurun_out/r04_bench_multi_3ctx.json
SPEC_BENCH_REHEARSE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 3 --log2-samples 26 --gather-steps 1 > gpurun_out/r04_rehearse_2ranks_gloo.json 2> gpurun_out/r04_rehearse.err || { tail -20 gpurun_out/r04_rehearse.err; exit 1; }
cat gpurun_out/r04_rehearse_2ranks_gloo.json
