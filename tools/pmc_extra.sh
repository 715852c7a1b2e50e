#!/bin/bash
# Development aid: further SQ / TA / TCP counters of one bench workload's kernels, one rocprofv3 --pmc pass per group.
#   tools/pmc_extra.sh <tag> <bench args...>   -> gpurun_out/pmcx_<tag>/summary.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcx_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
  "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH" \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" \
  "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE" \
  "SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" ; do   # (a TA_*_sum group aborted rocprofv3 on this pool: left out)
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/g$i.err || { echo "group $i failed: $grp"; tail -2 $OUT/g$i.err; }
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if any(x in k for x in ("team", "large_", "v2_kernel")):
        print(k, "(launches %d)" % len(next(iter(d.values()))))
        for c in sorted(d):
            print("   %-40s %.5g" % (c, sum(d[c]) / len(d[c])))
PY
cat $OUT/summary.txt
