#!/bin/bash
# memory-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the team kernel for fp32 lines and ring depths
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic3; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for cfg in "n65536f:" "n65536f:--opt large_ring=2" "cfg5:" "cfg5:--opt large_ring=3"; do
  w=${cfg%%:*}; o=${cfg#*:}; tag=$(echo "$w $o" | tr ' =-' '___')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/${tag}_$c -- python3 $ROOT/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline $o > $OUT/${tag}_$c.json 2> $OUT/${tag}_$c.err || tail -2 $OUT/${tag}_$c.err
  done
done
python3 - <<PY
import csv, glob, collections, os, json
out = "$OUT"
for d in sorted(set(p.rsplit("_", 2)[0] for p in glob.glob(out + "/*_SIZE"))):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob("%s_%s/*/*_counter_collection.csv" % (d, c)):
            for r in csv.DictReader(open(f)):
                if "large_team_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals.append(float(r["Counter_Value"]))
        tot[c] = sum(vals) / max(len(vals), 1)
    ms = None
    try:
        b = json.loads(open(d + "_WRITE_SIZE.json").read().strip().splitlines()[-1]); ms = b["roofline"]["kernel_ms"]; alg = b["roofline"]["bytes_per_line"] * b["roofline"]["lines_per_launch"]
    except Exception as e:
        alg = 0
    rd, wr = tot["FETCH_SIZE"] * 1024 * 2, tot["WRITE_SIZE"] * 1024
    print("%-40s read %.2f GB  write %.2f GB  total %.2f GB = %.2fx algorithmic (%.2f GB)  kernel %.3f ms under the profiler -> %.2f TB/s memory side"
          % (os.path.basename(d), rd / 1e9, wr / 1e9, (rd + wr) / 1e9, (rd + wr) / alg if alg else 0, alg / 1e9, ms or 0, (rd + wr) / (ms * 1e-3) / 1e12 if ms else 0))
PY
