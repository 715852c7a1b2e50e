#!/usr/bin/env python3
"""ISA lint: no vector-ALU instruction writes a data register of a buffer store wider than 8 bytes within the next
WAIT instruction slots (s_nop N counts N + 1).  Measured on gfx950 / ROCm 7.2 in spec_k_v3h.hip: hipcc restores a
borrowed half of the data tuple with a v_mov right behind `buffer_store_dwordx4 ... s<offset> offen`, and the last lanes
of the wave are stored with the new value; LLVM's hazard recogniser pads the pattern only for stores without a scalar
offset register.

    python tools/check_store_hazard.py <file.s> [...]      exit status 1 and one line per finding
"""
import re
import sys

WAIT = 2


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    findings, kern, ins = [], None, []
    for line in open(path):
        s = line.strip()
        if line.startswith("_Z") and ":" in line:
            kern = line.split(":")[0]
        if not kern or not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        ins.append((kern, s))
    for i, (k, s) in enumerate(ins):
        if not re.match(r"buffer_store_(dwordx[34]|format_xyzw?)\b", s):
            continue
        data = regs(re.split(r"[ ,]+", s)[1])
        slots, j = 0, i + 1
        while slots < WAIT and j < len(ins) and ins[j][0] == k:
            toks = re.split(r"[ ,]+", ins[j][1])
            if toks[0] == "s_nop":
                slots += int(toks[1], 0) + 1
            else:
                if toks[0].startswith("v_") and not toks[0].startswith("v_cmp") and len(toks) > 1 and regs(toks[1]) & data:
                    findings.append("%s: `%s` then, %d slot(s) later, `%s`" % (k[:80], s, slots + 1, ins[j][1]))
                slots += 1
            j += 1
    return findings


def main():
    bad = []
    for p in sys.argv[1:]:
        bad += check(p)
    for b in bad:
        print(b)
    print("%d finding(s)" % len(bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
