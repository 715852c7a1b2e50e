#!/bin/bash
# rocprofv3 evidence for one bench workload whose dominant kernel is NOT the first-by-total-time heuristics of
# summarize_profile.py (cfg5: the persistent team kernel; cfg4: Welch + finalize): kernel stats of the plain
# bench command, then FETCH_SIZE / WRITE_SIZE / L2 hit counters in separate --pmc passes.
#   tools/prof_team.sh <tag> <bench args...>      -> gpurun_out/prof_<tag>/
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-live-traffic "$@" > $OUT/bench_under_prof.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic "$@" > /dev/null 2> $OUT/pmc_$name.err || tail -3 $OUT/pmc_$name.err
done
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$OUT/stats/*/*_kernel_stats.csv")[0])))
print("kernel stats (plain bench command):")
for r in rows[:8]:
    print("  %-70s calls %5s  avg %10.4f ms  total %6.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("PMC means per launch:")
for k, d in agg.items():
    if any(x in k for x in ("team", "large_", "v2_kernel", "welch")):
        print("  %s: %s (launches %d)" % (k, {c: "%.5g" % (sum(v) / len(v)) for c, v in d.items()}, len(next(iter(d.values())))))
PY
