#!/usr/bin/env python3
"""Development aid: where the workgroups of the persistent large-N kernel (csrc/spec_k_team.hip) spend their time.
Needs the profiling variant of the library:
    python -m spectral_analyzer_amd.build --variant tPROF -- -DSPEC_TEAM_PROF
    SPEC_LIB_VARIANT=tPROF SPEC_TEAM_PROF_OUT=/tmp/prof.bin python bench.py --workload cfg5 --steps 3 --warmup 1 \
        --no-cpu-baseline --opt large_team=2        (the dump is that of the LAST launch)
    python tools/team_prof.py /tmp/prof.bin
Words per workgroup (shader-clock cycles of lane 0): 0 loop total, 1 counted wait at the top (column side: for the
previous line's stores; row side: for the tile), 2 wait for the prefetched rows / fall-back tile, 3 blocking ring
waits, 4 column side: first barrier, 5 number of blocking ring waits, 6 HW_REG_HW_ID | XCC_ID << 32,
7 role | team << 8 | lines << 32."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
a = a[a[:, 7] != 0]
role = (a[:, 7] & 0xFF).astype(int)
lines = (a[:, 7] >> 32).astype(np.float64)
for r, name in ((1, "column side"), (2, "row side")):
    s = a[role == r]
    if not len(s):
        continue
    n = lines[role == r]
    per = lambda k: (s[:, k] / n)
    print("%-11s %3d workgroups, %5.0f lines each: cycles per line  total %6.0f   top wait %6.0f   rows/tile wait %6.0f   "
          "ring wait %6.0f (%.2f blocking waits per line)   first barrier %6.0f"
          % (name, len(s), n.mean(), per(0).mean(), per(1).mean(), per(2).mean(), per(3).mean(), (s[:, 5] / n).mean(),
             per(4).mean()))

s = a[role == 2]
if len(s):
    n = lines[role == 2]
    names = ["top: wait + strips + pass 0 + store", "barrier 1", "hand-back + requests", "load + barrier 2", "pass 1 + store + barrier 3",
             "load + pass 2", "epilogue + output stores", "barrier 4"]
    print("row side phases (cycles per line, lane 0 of each workgroup):")
    for k, nm in enumerate(names):
        print("   %-40s %7.0f" % (nm, (s[:, 8 + k] / n).mean()))

# who shares a CU: HW_ID bits 8..15 (CU, shader array, shader engine) within an XCD
from collections import defaultdict
cu = defaultdict(list)
for row in a:
    cu[(int(row[6] >> 32) & 15, int(row[6] >> 8) & 0xFF)].append(int(row[7]) & 0xFF)
kinds = defaultdict(int)
for k, v in cu.items():
    kinds[tuple(sorted(v))] += 1
print("CUs by the roles of their workgroups (1 column, 2 row):", dict(kinds), " distinct (XCD, CU) keys:", len(cu))
