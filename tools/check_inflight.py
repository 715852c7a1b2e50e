#!/usr/bin/env python3
"""Static check of the software-pipelined loads of csrc/spec_k_team.hip.

The team kernel issues loads as inline assembly (so that hipcc does not wait for them with vmcnt(0)) and
waits for them itself (vm_wait<N>).  A load with a VGPR destination is dangerous there: the compiler believes
the value is present at once, and any copy it inserts between the load and the wait (phi resolution, live-range
splitting) copies what the load has not delivered yet -- the first pipelined version of the kernel produced
wrong lines exactly that way.  The kernel therefore uses LDS-DMA (global_load_lds: no register destination).
This script keeps it that way: it counts the LDS-DMA loads and, should a register-destination inline-assembly
load reappear, checks that no instruction touches its destination before the next inline-assembly s_waitcnt.  This script compiles the file to gfx950 assembly and scans every large_team_kernel:
a pass over the text in order that follows fall-through paths (the set of in-flight registers is kept across
labels and conditional branches and dropped at unconditional branches).  A lint, not a proof: an exact
data-flow over the structurised control flow reports paths no wave can take (role flags); the GPU parity tests
of every instantiation (tests/test_gpu_large.py) are the other half.  Exit code 1 and a listing when a register is touched while in flight.

    python tools/check_inflight.py [file.s]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "spectral_analyzer_amd", "csrc", "spec_k_team.hip")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return [int(m.group(1))] if m else []


def scan(text):
    """-> (kernels scanned, asm loads seen, violations [(kernel, line no, reg, issued at, text)])"""
    kernels, loads, dma, bad = 0, 0, 0, []
    name, inflight, inasm = None, {}, False
    for no, line in enumerate(text.split("\n"), 1):
        t = line.strip()
        m = re.match(r"(_ZN7specgpu\S*large_team_kernel\S*):", t)
        if m:
            name, inflight, inasm = m.group(1), {}, False
            kernels += 1
            continue
        if name is None:
            continue
        if t.startswith(".Lfunc_end"):
            name = None
            continue
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        if not t or t[0] in ";.":
            continue
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
        if inasm and t.startswith("global_load_lds"):  # LDS-DMA: no register destination, nothing to copy
            dma += 1
            continue
        if inasm and t.startswith("global_load"):
            loads += 1
            for r in regs(toks[0]):
                inflight[r] = no
            continue
        if t.startswith("s_waitcnt") and "vmcnt" in t and (inasm or "vmcnt(0)" in t):
            inflight = {}
            continue
        if t.startswith(("s_branch", "s_endpgm")):  # the text behind is not reached by falling through
            inflight = {}
            continue
        for tk in toks:
            for r in regs(tk):
                if r in inflight:
                    bad.append((name, no, r, inflight[r], t))
    return kernels, loads, dma, bad


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "team.s")
            # -DSPEC_TEAM_VARIANTS: the experiment geometries of lib/libspecgpu_teamvar.so are scanned with the product's
            subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast",
                                   "-DSPEC_TEAM_VARIANTS", "--cuda-device-only", "-S", SRC, "-o", out])
            text = open(out).read()
    kernels, loads, dma, bad = scan(text)
    print("kernels %d, LDS-DMA loads %d, inline-assembly loads with a register destination %d, "
          "in-flight registers touched %d" % (kernels, dma, loads, len(bad)))
    for b in bad[:40]:
        print("  %s line %d: v%d (load at line %d) in: %s" % (b[0][-48:], b[1], b[2], b[3], b[4]))
    return 1 if bad or kernels == 0 or dma == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
