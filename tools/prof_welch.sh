#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_welch
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/bench_other.py welch > $OUT/out.txt 2> $OUT/err.txt
grep -v amdgpu $OUT/out.txt
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$OUT/stats/*/*_kernel_trace.csv")[0])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:60], r["Grid_Size"], r["Workgroup_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    print("%-62s grid %9s wg %5s  n=%3d  median %.1f us" % (k[0], k[1], k[2], len(v), sorted(v)[len(v)//2]))
PY
