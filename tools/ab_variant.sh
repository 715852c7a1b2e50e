#!/bin/bash
# A/B of the product library against a named variant library ON ONE BOX (box-to-box spread is larger than most effects):
# alternating runs of bench.py per workload, kernel time by HIP events.   usage: tools/ab_variant.sh <variant> [workloads...]
V=$1; shift
W=${@:-cfg2 cfg3 cfg4 n16384 n32768f n65536f}
mkdir -p gpurun_out
for w in $W; do
  for rep in 1 2; do
    for lib in product $V; do
      if [ $lib = product ]; then unset SPEC_LIB_VARIANT; else export SPEC_LIB_VARIANT=$lib; fi
      timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --no-live-traffic > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "bench $w $lib failed"; tail -3 gpurun_out/ab_tmp.err; }
      python - "$w" "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
r = d["roofline"]
print("%-9s %-9s kernel_ms %.3f  frac %.3f  sclk %s  spot %s" % (sys.argv[1], sys.argv[2], r["kernel_ms"], r["frac"], r["clocks"]["sclk_mhz"]["median"], d["parity_spot_check"]["ok"]), flush=True)
PY
    done
  done
done
