#!/bin/bash
# rocprofv3 kernel statistics of the secondary benchmarks (burst chain, fused redraw, fp64 family, Welch).
#   tools/profile_secondary.sh   -> gpurun_out/prof_secondary/<name>_kernel_stats.csv
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_secondary
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() {  # tag script [args]
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $ROOT/tools/"$@" > $OUT/$tag.out 2> $OUT/$tag.err || tail -3 $OUT/$tag.err
  f=$(ls $OUT/$tag/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${tag}_kernel_stats.csv
}
run burst bench_burst.py
run render bench_render.py
run fp64 bench_cf64.py
run welch bench_other.py welch
ls $OUT/*_kernel_stats.csv
