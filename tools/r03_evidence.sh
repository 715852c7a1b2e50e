#!/bin/bash
# Round-3 evidence, part $1 (each part fits one gpurun call):
#   1  bench lines of every BASELINE workload (+ the fp32 65536-point lines), other sizes
#   2  rocprofv3 kernel stats + PMC passes for cfg2 and cfg3 (tools/profile.sh)
#   3  the same for cfg5 and cfg4 (tools/prof_team.sh)
#   4  further SQ counters of the cfg5 team kernel (tools/pmc_extra.sh), 2-rank gloo rehearsal of the N > 1 path
case "$1" in
1) bash tools/run_round_bench.sh r03
   timeout -k 10 300 python bench.py --workload n65536f --steps 20 --warmup 5 > gpurun_out/r03_bench_n65536f.json 2>/dev/null
   timeout -k 10 600 python tools/bench_other.py > gpurun_out/r03_other_configs.txt 2>&1; cat gpurun_out/r03_other_configs.txt
   timeout -k 10 300 python tools/bench_cf64.py > gpurun_out/r03_fp64_family.txt 2>&1; tail -12 gpurun_out/r03_fp64_family.txt ;;
2) bash tools/profile.sh r03cfg2 && bash tools/profile.sh r03cfg3 --workload cfg3 ;;
3) bash tools/prof_team.sh r03cfg5 --workload cfg5 --steps 10 --warmup 3 && bash tools/prof_team.sh r03cfg4 --workload cfg4 --steps 10 --warmup 3 ;;
4) bash tools/pmc_extra.sh cfg5 --workload cfg5; cat gpurun_out/pmcx_cfg5/summary.txt | head -60
   SPEC_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --log2-samples 26 > gpurun_out/r03_rehearse_2ranks_gloo.json 2> gpurun_out/r03_rehearse.err; echo "rehearse rc=$?"; tail -c 1500 gpurun_out/r03_rehearse_2ranks_gloo.json ;;
esac
