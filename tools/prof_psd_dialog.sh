ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_psd; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 $ROOT/tools/bench_psd_dialog.py > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv, glob
rows=list(csv.DictReader(open(glob.glob("$OUT/s/*/*_kernel_stats.csv")[0])))
for r in rows[:10]:
    print("%-70s calls %5s avg %9.1f us  max %9.1f us" % (r["Name"].replace("specgpu::","").replace("(anonymous namespace)::","").split("(")[0][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3))
PY
