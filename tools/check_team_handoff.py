#!/usr/bin/env python3
"""Static check of the inter-workgroup hand-off of csrc/spec_k_team.hip -- the contract its header comment argues,
asserted on the ISA so that a compiler upgrade that reorders either side fails a test instead of a line.

The team kernel hands line-sized tiles from "column" workgroups to "row" workgroups of the same XCD through that XCD's
L2, with RELAXED agent-scope counters and no release / acquire (an agent-scope release is `buffer_wbl2`: it would write
the L2-resident intermediate back to memory, the very traffic the kernel exists to avoid).  What makes that sound on
gfx950 (MI355X_MICROARCH.md, "inter-workgroup visibility", Valid forms):
  producer  every store of the handed-off bytes has left the CU (write-through L1, resident in L2) and has been WAITED
            for by the storing wave (s_waitcnt vmcnt) before one lane adds to the counter (agent-scope atomic);
  consumer  polls the counter with sc1 loads and reads the bytes with sc1 loads (L1 bypassed, L2 served), and hands the
            slot back (its own atomic add) only after those reads have been waited for.
The waits are COUNTED (vmcnt(N), N > 0: both sides keep stream traffic in flight across them), so the contract is:

  R1  the N youngest vector-memory operations in front of a hand-written counted wait are, on EVERY control-flow path,
      stream operations -- `nt` loads of the recording or `nt` stores of the output; never a slot store (plain
      global_store), a slot read or a counter poll (sc1 loads).  [vmcnt counts loads, stores and LDS-DMA together in
      issue order: "all but the N youngest are done".]
  R2  walking back from every counter add (global_atomic_add inside a loop), a vmcnt wait is met before any slot store
      on every path: no line is announced before its stores were waited for.  With R1 that wait really covers them.
  R3  every LDS-DMA load carries exactly one of `nt` (the recording) or `sc1` (slot, counter); every plain vector load
      of a counter or slot carries sc1; slot stores are plain (they must STAY in L2), output stores are `nt`.

A lint over the assembly text with a real control-flow graph (labels, conditional and unconditional branches), not a
proof of the hardware's behaviour: tests/test_gpu_large.py::test_team_and_two_launch_agree_on_every_value_at_full_size
is the other half.

    python tools/check_team_handoff.py [file.s]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "spectral_analyzer_amd", "csrc", "spec_k_team.hip")
VM = ("global_", "buffer_", "flat_", "scratch_")


class Ins:
    __slots__ = ("text", "asm", "op")

    def __init__(self, text, asm):
        self.text, self.asm, self.op = text, asm, text.split()[0]

    # classification ---------------------------------------------------------------------------------------
    def is_vm(self):
        return self.op.startswith(VM) and not self.op.startswith("buffer_inv") and not self.op.startswith("buffer_wbl2")

    def mods(self):
        return set(self.text.replace(",", " ").split()[1:]) & {"nt", "sc0", "sc1"}

    def is_slot_store(self):  # plain store of 8 / 16 bytes: the intermediate (output stores are nt)
        return self.op in ("global_store_dwordx4", "global_store_dwordx2") and not self.mods()

    def is_sc1_load(self):
        return ("_load" in self.op) and "sc1" in self.mods()

    def vmcnt(self):
        m = re.search(r"vmcnt\((\d+)\)", self.text) if self.op == "s_waitcnt" else None
        return int(m.group(1)) if m else None


def kernels(text):
    """-> {name: [Ins]} for every large_team_kernel in the assembly text"""
    out, name, asm = {}, None, False
    for line in text.split("\n"):
        t = line.strip()
        m = re.match(r"(_ZN7specgpu\S*large_team_kernel\S*):", t)
        if m:
            name, asm = m.group(1), False
            out[name] = []
            continue
        if name is None:
            continue
        if t.startswith(".Lfunc_end"):
            name = None
            continue
        if t.startswith(";;#ASMSTART"):
            asm = True
            continue
        if t.startswith(";;#ASMEND"):
            asm = False
            continue
        if not t or t[0] == ";":
            continue
        if t[0] == "." and not re.match(r"\.LBB\d+_\d+:", t):
            continue
        out[name].append(Ins(t.split(";")[0].strip(), asm))
    return out


def cfg(ins):
    """predecessor lists over instruction indices (labels are nodes of their own)"""
    label = {i.text[:-1]: k for k, i in enumerate(ins) if i.text.endswith(":")}
    pred = [[] for _ in ins]
    for k, i in enumerate(ins):
        fall = True
        if i.op.startswith(("s_branch", "s_cbranch")):
            tgt = i.text.split()[-1]
            if tgt in label:
                pred[label[tgt]].append(k)
            fall = not i.op.startswith("s_branch")
        if i.op in ("s_endpgm", "s_setpc_b64"):
            fall = False
        if fall and k + 1 < len(ins):
            pred[k + 1].append(k)
    return pred


def youngest_before(ins, pred, at, n):
    """All vector-memory instructions that can be among the n youngest in front of instruction `at` on some path.
    A path ends at a vmcnt(0) wait (everything older is done) and at the kernel's entry."""
    found, seen, stack = set(), set(), [(p, n) for p in pred[at]]
    while stack:
        k, left = stack.pop()
        if (k, left) in seen:
            continue
        seen.add((k, left))
        i = ins[k]
        if i.vmcnt() == 0:
            continue
        if i.is_vm():
            found.add(k)
            left -= 1
            if left == 0:
                continue
        stack.extend((p, left) for p in pred[k])
    return found


def stores_before_a_wait(ins, pred, at):
    """slot stores reachable backwards from `at` without crossing a vmcnt wait"""
    bad, seen, stack = set(), set(), list(pred[at])
    while stack:
        k = stack.pop()
        if k in seen:
            continue
        seen.add(k)
        i = ins[k]
        if i.vmcnt() is not None:
            continue
        if i.is_slot_store():
            bad.add(k)
            continue
        stack.extend(pred[k])
    return bad


def in_loop(pred, k):
    """is instruction k on a cycle of the control-flow graph?"""
    seen, stack = set(), list(pred[k])
    while stack:
        j = stack.pop()
        if j == k:
            return True
        if j in seen:
            continue
        seen.add(j)
        stack.extend(pred[j])
    return False


def check(text):
    report, counts = [], dict(kernels=0, counted_waits=0, adds=0, dma=0, slot_stores=0)
    for name, ins in kernels(text).items():
        counts["kernels"] += 1
        pred = cfg(ins)
        short = name[-44:]
        for k, i in enumerate(ins):
            n = i.vmcnt()
            if i.asm and n:  # R1
                counts["counted_waits"] += 1
                for j in sorted(youngest_before(ins, pred, k, n)):
                    if ins[j].is_slot_store() or ins[j].is_sc1_load() or "atomic" in ins[j].op:
                        report.append("R1 %s: `%s` (instr %d) may still be in flight at `%s` (instr %d)" % (short, ins[j].text, j, i.text, k))
            if i.op.startswith("global_atomic_add") and in_loop(pred, k):  # R2
                counts["adds"] += 1
                for j in sorted(stores_before_a_wait(ins, pred, k)):
                    report.append("R2 %s: slot store `%s` (instr %d) reaches the counter add at instr %d without a vmcnt wait" % (short, ins[j].text, j, k))
            if i.op.startswith("global_load_lds"):  # R3
                counts["dma"] += 1
                if len(i.mods() & {"nt", "sc1"}) != 1:
                    report.append("R3 %s: LDS-DMA load `%s` carries neither / both of nt, sc1" % (short, i.text))
            if i.is_slot_store():
                counts["slot_stores"] += 1
            if i.op.startswith("global_store") and i.op not in ("global_store_dword",) and i.mods() - {"nt"}:
                report.append("R3 %s: store `%s` is neither plain (slot) nor nt (output)" % (short, i.text))
            if i.op.startswith("buffer_load") and "sc1" not in i.mods() and in_loop(pred, k) and i.op != "buffer_load_dword":
                pass  # raw-buffer recording loads of the plain forms (nt); slot reads through ld_slot carry sc1 -- checked on the GPU
    return counts, report


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "team.s")
            subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast",
                                   "-DSPEC_TEAM_VARIANTS", "--cuda-device-only", "-S", SRC, "-o", out])
            text = open(out).read()
    counts, report = check(text)
    print("kernels %(kernels)d, hand-written counted waits %(counted_waits)d, counter adds in loops %(adds)d, "
          "LDS-DMA loads %(dma)d, slot stores %(slot_stores)d" % counts, "violations %d" % len(report))
    for r in report[:60]:
        print("  " + r)
    ok = not report and counts["kernels"] and counts["counted_waits"] and counts["adds"] and counts["slot_stores"]
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
