#!/bin/bash
# Round-4 evidence, part $1 (each part fits one gpurun call):
#   1  bench lines of every BASELINE workload (+ fp32 65536- and 32768-point lines), other sizes, fp64 family, v2h timings
#   2  rocprofv3 kernel stats + PMC passes (full counter set: tools/profile.sh) for cfg2, cfg3, the new 32768-point kernel
#   3  the same for cfg4 and 16384-point lines (issue table), prof_team for cfg5 (traffic)
R=r04
case "$1" in
1) bash tools/run_round_bench.sh $R
   timeout -k 10 300 python bench.py --workload n65536f --steps 20 --warmup 5 > gpurun_out/${R}_bench_n65536f.json 2>/dev/null
   timeout -k 10 300 python bench.py --workload n32768f --steps 20 --warmup 5 > gpurun_out/${R}_bench_n32768f.json 2>/dev/null; cat gpurun_out/${R}_bench_n32768f.json
   timeout -k 10 600 python tools/bench_other.py > gpurun_out/${R}_other_configs.txt 2>&1; cat gpurun_out/${R}_other_configs.txt
   timeout -k 10 300 python tools/bench_cf64.py > gpurun_out/${R}_fp64_family.txt 2>&1; tail -12 gpurun_out/${R}_fp64_family.txt
   timeout -k 10 300 python tools/bench_v2h.py base lpw > gpurun_out/${R}_v2h.txt 2>&1; cat gpurun_out/${R}_v2h.txt ;;
2) bash tools/profile.sh ${R}cfg2 && bash tools/profile.sh ${R}cfg3 --workload cfg3 && bash tools/profile.sh ${R}n32768f --workload n32768f --steps 20 --warmup 5 ;;
3) bash tools/profile.sh ${R}cfg4 --workload cfg4 --steps 10 --warmup 3 && bash tools/profile.sh ${R}n16384 --workload n16384 --steps 20 --warmup 5 && bash tools/prof_team.sh ${R}cfg5 --workload cfg5 --steps 10 --warmup 3 ;;
esac
