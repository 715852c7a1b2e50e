#!/bin/bash
# kernel-level timing of the large-N path (development tool)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_cfg5
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/bench_other.py cfg5 > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt | grep -v amdgpu
cat $OUT/stats/*/*_kernel_stats.csv | cut -c1-200 | head -8
