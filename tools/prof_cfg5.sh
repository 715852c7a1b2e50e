#!/bin/bash
# kernel-level timing and HBM traffic of the large-N path (development tool)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_cfg5
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/bench_other.py cfg5only > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt | grep -v amdgpu
cat $OUT/stats/*/*_kernel_stats.csv | cut -c1-160 | head -6
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '+')
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/tools/bench_other.py cfg5only > /dev/null 2> $OUT/pmc_$name.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "large_" in k:
            agg["cols" if "cols" in k else "rows"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
