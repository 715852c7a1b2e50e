#!/bin/bash
# Round 5, item 1: the 16384-point Welch plan with one workgroup-wide exchange (spec_v2.h Plan2<214>, "welch_rows") beside the
# 32 x 32 x 16 plan: parity cases, cfg4 timed both ways, the per-wave time line of both (variant library v2stamp).
mkdir -p gpurun_out
O=gpurun_out/r05_rows.txt
: > $O
timeout -k 10 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_large.py -x -q -m gpu -k "welch or cfg4 or psd" > gpurun_out/r05_rows_pytest.txt 2>&1
echo "pytest (welch cases, welch_rows default): rc $? -- $(tail -1 gpurun_out/r05_rows_pytest.txt)" >> $O
[ "$(tail -1 gpurun_out/r05_rows_pytest.txt | grep -c failed)" = 0 ] || { tail -40 gpurun_out/r05_rows_pytest.txt; exit 1; }
for rows in 0 1; do
  for rep in 1 2; do
    echo "cfg4 welch_rows=$rows: $(timeout -k 10 300 python bench.py --workload cfg4 --steps 20 --warmup 5 --no-cpu-baseline --opt welch_rows=$rows 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g seg/s  kernel %.3f ms  frac %.3f  parity_ok=%s" % (d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["parity_spot_check"]["ok"]))')" >> $O
  done
done
for rows in 0 1; do
  SPEC_LIB_VARIANT=v2stamp timeout -k 10 300 python tools/v2_timeline.py $rows 100 0 >> $O 2>&1
  SPEC_LIB_VARIANT=v2stamp timeout -k 10 300 python tools/v2_timeline.py $rows 100 5 | tail -1 >> $O 2>&1
done
cat $O
