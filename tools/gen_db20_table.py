#!/usr/bin/env python3
"""Generates DB20_TAB of csrc/spec_fft.h (the table logarithm of the fp64 dB epilogue) and checks the whole
expression, evaluated operation by operation in round-to-nearest fp64 (mpmath at 53 bits, fused multiply-adds
as single roundings), against 60-digit arithmetic.

    python tools/gen_db20_table.py            # prints the table (C initialiser) and the measured error
    python tools/gen_db20_table.py --check    # exit code 1 if spec_fft.h holds a different table

ln p for p = m 2^e, m in [1, 2) cut into NI = 128 intervals: ln m = ln(m inv_i) - ln(inv_i) with inv_i =
fp64(1 / centre of interval i) and -ln(inv_i) tabulated for that ROUNDED inv_i (an identity, not an approximation);
|m inv_i - 1| <= 2^-8, so ln(1 + r) = r - r^2/2 + r^3/3 - r^4/4 + r^5/5 leaves r^6/6 < 6e-16."""
import os
import re
import struct
import sys

import mpmath as mp

NI = 128
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def f64(x):
    """round an mpf to the nearest double, as a Python float"""
    with mp.workprec(53):
        return float(+mp.mpf(x))


def table():
    mp.mp.prec = 300
    t = []
    for i in range(NI):
        inv = f64(1 / (1 + (mp.mpf(i) + mp.mpf(1) / 2) / NI))
        t += [inv, f64(-mp.log(mp.mpf(inv)))]
    return t


def c_text(t):
    rows = []
    for k in range(0, len(t), 4):
        rows.append("    " + " ".join("%s," % float.hex(v) for v in t[k:k + 4]))
    return "\n".join(rows)


def fma(a, b, c):
    return f64(mp.mpf(a) * mp.mpf(b) + mp.mpf(c))


def ln_tab(v, t):
    bits = struct.unpack("<Q", struct.pack("<d", v))[0]
    hi = bits >> 32
    e = float((hi >> 20) - 1023)
    m = struct.unpack("<d", struct.pack("<Q", (bits & 0x000FFFFFFFFFFFFF) | 0x3FF0000000000000))[0]
    i = (hi >> 13) & (NI - 1)
    inv, li = t[2 * i], t[2 * i + 1]
    r = fma(m, inv, -1.0)
    q = 1.0 / 5
    q = fma(q, r, -1.0 / 4)
    q = fma(q, r, 1.0 / 3)
    q = fma(q, r, -0.5)
    s = fma(f64(mp.mpf(r) * mp.mpf(r)), q, r)
    return fma(e, float.fromhex("0x1.62e42fefa39efp-1"), f64(mp.mpf(li) + mp.mpf(s)))


def db20_fast(x, y, t):
    """the series form for |X|^2 in (2^-13, 2^996), an exact 1 / |X| standing in for the hardware estimate"""
    k10 = float.fromhex("0x1.15f2ced384f29p+2")
    p = fma(x, x, f64(mp.mpf(y) * mp.mpf(y)))
    rs = f64(1 / mp.sqrt(mp.mpf(p)))
    return fma(ln_tab(p, t), k10, f64(mp.mpf(float.fromhex("0x1.dd8307784b277p-31")) * mp.mpf(rs)))


def main():
    t = table()
    if "--check" in sys.argv:
        src = open(os.path.join(ROOT, "spectral_analyzer_amd", "csrc", "spec_fft.h")).read()
        body = src[src.index("DB20_TAB[%d] = {" % (2 * NI)):]
        body = body[:body.index("};")]
        have = [float.fromhex(h) for h in re.findall(r"-?0x[0-9a-f.]+p[+-]?\d+", body)]
        ok = have == t
        print("spec_fft.h table %s (%d entries)" % ("matches" if ok else "DIFFERS", len(have)))
        return 0 if ok else 1
    print(c_text(t))
    import random
    rnd = random.Random(5)
    worst = 0.0
    n = int(os.environ.get("N", "20000"))
    for _ in range(n):
        mag = 10.0 ** rnd.uniform(-1.95, 18)
        ph = rnd.uniform(0, 6.283185307179586)
        x, y = mag * mp.cos(ph), mag * mp.sin(ph)
        x, y = f64(x), f64(y)
        got = db20_fast(x, y, t)
        want = 20 * mp.log10(mp.sqrt(mp.mpf(x) ** 2 + mp.mpf(y) ** 2) + mp.mpf("1e-10"))
        worst = max(worst, abs(float(mp.mpf(got) - want)))
    print("// %d magnitudes in 1.1e-2 ... 1e18: max |error| %.3g dB" % (n, worst), file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
