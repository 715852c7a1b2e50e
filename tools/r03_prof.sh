#!/bin/bash
# phase profile of the team kernel (variant tPROF) for cfg5 and the fp32 65536-point lines
mkdir -p gpurun_out/prof3
for w in cfg5 n65536f; do
  SPEC_LIB_VARIANT=tPROF SPEC_TEAM_PROF_OUT=/tmp/prof_$w.bin timeout -k 10 200 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --opt large_team=2 > gpurun_out/prof3/$w.json 2> gpurun_out/prof3/$w.err
  echo "== $w"; python tools/team_prof.py /tmp/prof_$w.bin | tee gpurun_out/prof3/$w.txt; python tools/team_trace.py /tmp/prof_$w.bin | tee gpurun_out/prof3/${w}_trace.txt
done
