#!/usr/bin/env python3
"""Static instruction budget of a kernel's line loop, from the assembly hipcc emits (-save-temps): what the vector
ALU, the LDS and the memory pipe are asked to do per line, by kind -- the table DESIGN.md sets beside the counters
(SQ_INSTS_VALU per wave and line) when it says a kernel is "issue-bound at the minimum instruction count".

    python tools/isa_budget.py <file.s> <kernel-name-substring> [--lines-per-iteration K]

The line loop is the backward branch that spans the most instructions.  Inside it, basic blocks that hold a square
root (v_sqrt_f32: the exact form of the dB epilogue, taken only when a bin leaves the fast range) are counted
separately as "cold" (so are the per-bin blocks of that chain: exactly one v_log_f32 each).  --lines-per-iteration: 2 for the ping-pong body of the cf32 50 %-overlap kernels (two lines per
trip), 1 otherwise.
"""
import collections
import re
import sys


def classify(op: str) -> str:
    if op.startswith("v_pk_add"):
        return "valu: packed add/sub (butterflies)"
    if op.startswith("v_pk_mul"):
        return "valu: packed mul (twiddle first half, constant rotations, window)"
    if op.startswith("v_pk_fma"):
        return "valu: packed fma (twiddle second half, +-i and h-scaled butterflies, dB scale)"
    if op.startswith("v_cvt") or op.startswith("v_perm") or op.startswith("v_bfe") or op.startswith("v_bfi"):
        return "valu: decode (conversions, byte swaps)"
    if op.startswith(("v_log", "v_exp", "v_sqrt", "v_rsq", "v_rcp")):
        return "valu: transcendental (epilogue)"
    if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mac_f32", "v_mad_f32",
                      "v_fma_f64", "v_mul_f64", "v_add_f64")):
        return "valu: scalar-lane fp math (|X|^2, sums, window)"
    if op.startswith(("v_min", "v_max", "v_cmp", "v_cndmask", "v_med3")):
        return "valu: range test / selects (epilogue)"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
        return "valu: moves (overlap shift, copies)"
    if op.startswith("v_"):
        return "valu: integer / addressing / other"
    if op.startswith("ds_"):
        return "lds: " + ("writes" if "write" in op or "store" in op else "reads")
    if op.startswith(("buffer_load", "global_load", "flat_load")):
        return "vmem: loads"
    if op.startswith(("buffer_store", "global_store", "flat_store")):
        return "vmem: stores"
    if op.startswith("scratch_"):
        return "vmem: scratch (spills)"
    if op == "s_barrier":
        return "sync: s_barrier"
    if op == "s_waitcnt":
        return "sync: s_waitcnt"
    if op.startswith("s_nop"):
        return "salu: s_nop (packed-math wait states)"
    if op.startswith("s_"):
        return "salu: other"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    per = 1
    if "--lines-per-iteration" in sys.argv:
        per = int(sys.argv[sys.argv.index("--lines-per-iteration") + 1])
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    ins, labels = [], {}
    for i in range(start + 1, end + 1):
        l = lines[i]
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
            toks = l.strip().split()
            ins.append((toks[0], l.strip()))
    # the backward branch spanning the most instructions
    best = None
    for j, (op, text) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = text.split()[-1]
            if tgt in labels and labels[tgt] <= j:
                if best is None or j - labels[tgt] > best[1] - best[0]:
                    best = (labels[tgt], j)
    if best is None:
        print("no loop found")
        return
    a, b = best
    # basic blocks inside the loop
    bounds = sorted(set([a, b + 1] + [v for v in labels.values() if a < v <= b] +
                        [j + 1 for j in range(a, b + 1) if ins[j][0].startswith(("s_cbranch", "s_branch"))]))
    hot, cold = collections.Counter(), collections.Counter()
    for s, e in zip(bounds[:-1], bounds[1:]):
        blk = ins[s:e]
        # the exact dB form is a chain of per-bin blocks (one v_log_f32 each, a v_sqrt_f32 behind a branch); the fast form
        # is ONE block with all the thread's logarithms
        n_log = sum(1 for op, _ in blk if op.startswith("v_log"))
        is_cold = any(op.startswith("v_sqrt") for op, _ in blk) or n_log == 1
        for op, _ in blk:
            (cold if is_cold else hot)[classify(op)] += 1
    name = lines[start].split(":")[0]
    print("kernel %s" % name[:110])
    print("line loop: %d instructions, %d line(s) per trip; per LINE and wave:" % (b - a + 1, per))
    tot = collections.Counter()
    for k in sorted(set(hot) | set(cold)):
        print("  %-82s %7.1f   (+ %5.1f in the cold exact-dB blocks)" % (k, hot[k] / per, cold[k] / per))
        tot[k.split(":")[0]] += hot[k]
    print("  ---")
    for k in sorted(tot):
        print("  %-82s %7.1f" % ("total " + k + " on the hot path", tot[k] / per))


if __name__ == "__main__":
    main()
