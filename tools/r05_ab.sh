#!/bin/bash
# Round 5: barrier placement / fused radix-32 stores A/B over the workloads whose kernels include spec_v2.h's v2_fft.
#   usage: tools/r05_ab.sh "<variants>" "<workloads>"      ("" = the product library)
mkdir -p gpurun_out
O=gpurun_out/r05_ab.txt
VARS=${1:-"product v2late v2nofuse"}
WLS=${2:-"cfg4 n16384 n32768f cfg2 cfg3"}
line() { python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.4g /s  kernel %.3f ms  frac %.3f  parity_ok=%s" % (d["value"], r["kernel_ms"], r["frac"], d["parity_spot_check"]["ok"]))'; }
for w in $WLS; do
  for v in $VARS; do
    if [ $v = product ]; then unset SPEC_LIB_VARIANT; else export SPEC_LIB_VARIANT=$v; fi
    for opt in ${OPTS:-none}; do
      o=""; [ $opt != none ] && o="--opt $opt"
      echo "$w $v $opt: $(timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic $o 2>/dev/null | line)" | tee -a $O
    done
  done
done
