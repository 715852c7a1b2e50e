#!/bin/bash
# Round evidence in one go (on the GPU box): bench lines for every BASELINE workload, then the rocprofv3 passes of
# the two workloads whose kernels changed this round.  Outputs under gpurun_out/; tools/summarize_profile.py turns
# the profile directories into profiles/<round>_*.
#   usage: tools/run_round_bench.sh <round-tag>
R=${1:-r05}
mkdir -p gpurun_out
for w in cfg2 cfg3 cfg1; do
  timeout -k 10 400 python bench.py --workload $w > gpurun_out/${R}_bench_$w.json 2> gpurun_out/${R}_bench_$w.err || echo "bench $w failed"
done
timeout -k 10 400 python bench.py --workload cfg5 --steps 20 --warmup 5 > gpurun_out/${R}_bench_cfg5.json 2> gpurun_out/${R}_bench_cfg5.err || echo "bench cfg5 failed"
timeout -k 10 400 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --opt large_team=0 > gpurun_out/${R}_bench_cfg5_two_launch.json 2>/dev/null || echo "bench cfg5 (two-launch) failed"
timeout -k 10 400 python bench.py --workload cfg4 --steps 20 --warmup 5 > gpurun_out/${R}_bench_cfg4.json 2> gpurun_out/${R}_bench_cfg4.err || echo "bench cfg4 failed"
timeout -k 10 400 python bench.py --workload cfg4 --steps 200 --warmup 20 --n-psd 1 --no-cpu-baseline > gpurun_out/${R}_bench_cfg4_one_psd.json 2>/dev/null || echo "bench cfg4 (1 PSD) failed"
cat gpurun_out/${R}_bench_*.json
