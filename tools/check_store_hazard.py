#!/usr/bin/env python3
"""ISA lint: no vector-ALU instruction writes a data register of a buffer store wider than 8 bytes within the next
WAIT instruction slots (s_nop N counts N + 1).  Measured on gfx950 / ROCm 7.2 in spec_k_v3h.hip: hipcc restores a
borrowed half of the data tuple with a v_mov right behind `buffer_store_dwordx4 ... s<offset> offen`, and the last lanes
of the wave are stored with the new value; LLVM's hazard recogniser pads the pattern only for stores without a scalar
offset register.

    python tools/check_store_hazard.py <file.s> [...]      exit status 1 and one line per finding
The check itself lives in spectral_analyzer_amd/build.py (store_hazard_findings): every build of every library -- product and
variants -- runs it on the device assembly of every translation unit and fails on a finding.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spectral_analyzer_amd.build import store_hazard_findings as check  # noqa: E402  (ONE implementation: the build runs it on every unit)


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
