#!/usr/bin/env python3
"""Development aid: live vector registers along the straight-line text of one kernel's assembly (hipcc -S), printed per
barrier-delimited region -- where a kernel that spills runs out of registers.  Branches are ignored (the text order of
the hot loop is taken as the path), so cold blocks inside the loop inflate nothing but their own lines.
    python tools/vgpr_pressure.py file.s <kernel-name-substring> [first_line last_line]
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
STORE = re.compile(r"^(ds_write|ds_store|buffer_store|global_store|scratch_store|flat_store|s_|buffer_wb|;)")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    op, _, rest = line.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    if STORE.match(op) or op.startswith("v_cmp") and not op.startswith("v_cmpx") or op in ("s_waitcnt", "s_barrier", "s_nop"):
        return op, set(), set().union(*[regs(o) for o in ops]) if ops else set()
    if op.startswith("v_cmp"):
        return op, set(), set().union(*[regs(o) for o in ops])
    d = regs(ops[0]) if ops else set()
    u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
    if op.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c")):
        u |= d
    return op, d, u


def main():
    text = open(sys.argv[1]).read().splitlines()
    start = next(i for i, l in enumerate(text) if sys.argv[2] in l and re.match(r"^[A-Za-z_][\w.$]*:", l))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    body = text[start:end]
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(body))
    ins = [(i, parse(body[i])) for i in range(lo, hi)]
    ins = [(i, p) for i, p in ins if p]
    live, counts = set(), {}
    # two backward sweeps over the range: the second sees what the loop carries
    for _ in range(2):
        for i, (op, d, u) in reversed(ins):
            live -= d
            live |= u
            counts[i] = len(live)
    region, peak, peak_at, first = 0, 0, lo, lo
    for i, (op, d, u) in ins:
        if counts[i] > peak:
            peak, peak_at = counts[i], i
        if op == "s_barrier" or i == ins[-1][0]:
            print("lines %5d-%5d (region %d, ends with %s): peak %3d live VGPRs at line %d: %s" % (first, i, region, op, peak, peak_at, body[peak_at].strip()[:60]))
            region, peak, first = region + 1, 0, i + 1


if __name__ == "__main__":
    main()
