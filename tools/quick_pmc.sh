#!/bin/bash
# Development aid: HBM-side traffic of a bench workload's dominant kernel in two short rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE; the guide's gfx950 corrections applied by tools/quick_pmc.py).
#   tools/quick_pmc.sh <tag> <bench args...>        e.g.  tools/quick_pmc.sh pair --workload n65536f
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/qpmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic "$@" > $OUT/$name.json 2> $OUT/$name.err || tail -3 $OUT/$name.err
done
python3 $ROOT/tools/quick_pmc.py $OUT
