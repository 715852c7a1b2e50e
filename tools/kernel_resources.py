#!/usr/bin/env python3
"""Development aid: registers, scratch and spills of every kernel of one translation unit, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.
    python tools/kernel_resources.py spectral_analyzer_amd/csrc/spec_k_v2w.hip [filter-regex] [-- extra hipcc flags]
"""
import re
import subprocess
import sys

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spectral_analyzer_amd import build as b  # noqa: E402


def resources(src, extra=()):
    cmd = [b._hipcc(), *[f for f in b.FLAGS if f != "--offload-compress"], *extra, "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr[-3000:])
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
        if not m:
            continue
        k, _, v = m.group(1).partition(":")
        if k.strip() == "Function Name":
            cur = out.setdefault(v.strip(), {})
        elif cur is not None:
            cur[k.strip()] = v.strip()
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


if __name__ == "__main__":
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    res = resources(args[0], extra)
    pat = re.compile(args[1]) if len(args) > 1 else None
    names = list(res)
    for n, d in zip(names, demangle(names)):
        d = re.sub(r"specgpu::\(anonymous namespace\)::", "", d).replace("(specgpu::(anonymous namespace)::", "(")
        if pat and not pat.search(d):
            continue
        r = res[n]
        print("%-70s VGPR %3s AGPR %3s scratch %4s B/lane  spilled VGPRs %3s  LDS %s" % (
            d[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
