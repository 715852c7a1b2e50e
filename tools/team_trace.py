#!/usr/bin/env python3
"""Development aid: the hand-offs of ONE team of the persistent large-N kernel on a common clock (variant tPROF).
    SPEC_LIB_VARIANT=tPROF SPEC_TEAM_PROF_OUT=/tmp/prof.bin python bench.py --workload cfg5 --steps 3 --warmup 1 \
        --no-cpu-baseline --opt large_team=2 ; python tools/team_trace.py /tmp/prof.bin [NT]
Events (wall clock, 10 ns ticks) of lane 0 of every member workgroup of team 0 for 32 consecutive lines:
 column side 0 line start, 1 previous line's stores waited for (announce), 2 behind the first barrier, 3 arithmetic
 done, 4 slot free (after a blocking wait, if any), 5 stores and next requests issued;
 row side 0 line start, 1 tile landed, 2 hand-back = decision about the next tile (7 set: it is being prefetched), 4
 epilogue and stores done, 5 after the blocking wait, 6 after the exposed tile read."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tr = raw[16384:16384 + 64 * 32 * 8].reshape(64, 32, 8).astype(np.int64)
col, row = tr[:nt], tr[nt:2 * nt]
t0 = col[:, :, 0][col[:, :, 0] > 0].min()
us = lambda x: (x - t0) / 100.0
L = range(4, 14)
print("times in us relative to the first traced event; per line: min .. max over the %d workgroups of a side" % nt)
for i in L:
    c, r = col[:, i, :], row[:, i, :]
    def mm(x):
        x = x[x > 0]
        return "%7.2f..%7.2f" % (us(x.min()), us(x.max())) if len(x) else "      -        "
    print("line %2d  COL start %s announce(prev) %s arith done %s slot free %s stored %s" % (i, mm(c[:, 0]), mm(c[:, 1]), mm(c[:, 3]), mm(c[:, 4]), mm(c[:, 5])))
    print("         ROW start %s tile in %s hand-back %s (prefetch %2d/%d) done %s unblocked %s tile read %s"
          % (mm(r[:, 0]), mm(r[:, 1]), mm(r[:, 2]), int((r[:, 7] > 0).sum()), nt, mm(r[:, 4]), mm(r[:, 5]), mm(r[:, 6])))
per = (col[:, 13, 0] - col[:, 3, 0]).mean() / 10 / 100.0
print("mean line period %.2f us" % per)
