#!/bin/bash
# Collect the rocprofv3 evidence for the bench workload on the GPU box.
#   tools/profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,pmc_*}; copy summaries into profiles/.
set -e
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic $@"
# the stats pass runs the SAME command as the plain bench (default steps/warmup), so that the
# kernel's average duration can be compared with bench.py's HIP-event time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-live-traffic $@ > $OUT/bench_under_prof.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err || { tail -5 $OUT/pmc_$name.err; }
done
find $OUT -name "*.csv" | head -30
