#!/bin/bash
# Round-5 evidence, part $1 (each part fits one gpurun call):
#   1  bench lines of every workload (clocks inside), other sizes, fp64 family, 32768- and 65536-point sweeps
#   2  rocprofv3 kernel stats + PMC passes (tools/profile.sh) for cfg2, cfg3, n65536f (the new paired kernel)
#   3  the same for cfg4, n16384, n32768f
#   4  prof_team for cfg5, profile for n16384d
R=r05
case "$1" in
1) bash tools/run_round_bench.sh $R
   for w in n65536f n32768f n16384 n16384d; do
     timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/${R}_bench_$w.json 2>/dev/null; cat gpurun_out/${R}_bench_$w.json
   done
   timeout -k 10 600 python tools/bench_other.py > gpurun_out/${R}_other_configs.txt 2>&1; cat gpurun_out/${R}_other_configs.txt
   timeout -k 10 300 python tools/bench_cf64.py > gpurun_out/${R}_fp64_family.txt 2>&1; tail -12 gpurun_out/${R}_fp64_family.txt
   timeout -k 10 300 python tools/bench_v2h.py base > gpurun_out/${R}_v2h.txt 2>&1; cat gpurun_out/${R}_v2h.txt
   timeout -k 10 600 python tools/bench_v2q.py base 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_v2q.txt; cat gpurun_out/${R}_v2q.txt ;;
2) bash tools/profile.sh ${R}cfg2 && bash tools/profile.sh ${R}cfg3 --workload cfg3 && bash tools/profile.sh ${R}n65536f --workload n65536f --steps 20 --warmup 5 ;;
3) bash tools/profile.sh ${R}cfg4 --workload cfg4 --steps 10 --warmup 3 && bash tools/profile.sh ${R}n16384 --workload n16384 --steps 20 --warmup 5 && bash tools/profile.sh ${R}n32768f --workload n32768f --steps 20 --warmup 5 ;;
4) bash tools/prof_team.sh ${R}cfg5 --workload cfg5 --steps 10 --warmup 3 && bash tools/profile.sh ${R}n16384d --workload n16384d --steps 20 --warmup 5 ;;
esac
