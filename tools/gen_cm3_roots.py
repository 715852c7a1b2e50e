#!/usr/bin/env python3
"""Generates the per-stage roots of unity that commons-math3 3.6.1's FastFourierTransformer holds as the
hexadecimal double literals W_SUB_N_R[k], W_SUB_N_I[k], k = 0 .. 62:

    W_SUB_N_R[k] =  cos(2 pi / 2^k),   W_SUB_N_I[k] = -sin(2 pi / 2^k)

evaluated AT THE DOUBLE ARGUMENT fl(2 pi) / 2^k (the library's literals are the doubles nearest to the cosine /
sine of that double, which is why W_SUB_N_R[2] is 0x1.1a62633145c07p-54 and not 0, and W_SUB_N_I[0] is
0x1.1a62633145c07p-52 and not 0).  The jar is not in the build image; the values are recomputed here with
60-digit arithmetic (mpmath) and rounded once.  Known literals of the published source, kept as a check:
    W_SUB_N_R[0..3] = 0x1.0p0, -0x1.0p0, 0x1.1a62633145c07p-54, 0x1.6a09e667f3bcdp-1
    W_SUB_N_I[0..3] = 0x1.1a62633145c07p-52, -0x1.1a62633145c07p-53, -0x1.0p0, -0x1.6a09e667f3bccp-1

    python tools/gen_cm3_roots.py            # prints the C initialisers pasted into oracle/spec_oracle.c
"""
import math

import mpmath

mpmath.mp.prec = 240
TWO_PI = mpmath.mpf(2.0 * math.pi)  # the double fl(2 pi), exactly


def roots():
    wr, wi = [], []
    for k in range(63):
        a = TWO_PI / (mpmath.mpf(2) ** k)  # exact: a power-of-two division of a double
        wr.append(float(mpmath.cos(a)))
        wi.append(float(-mpmath.sin(a)))
    return wr, wi


def main():
    wr, wi = roots()
    known_r = [float.fromhex(s) for s in ("0x1.0p0", "-0x1.0p0", "0x1.1a62633145c07p-54", "0x1.6a09e667f3bcdp-1")]
    known_i = [float.fromhex(s) for s in ("0x1.1a62633145c07p-52", "-0x1.1a62633145c07p-53", "-0x1.0p0", "-0x1.6a09e667f3bccp-1")]
    assert wr[:4] == known_r, [x.hex() for x in wr[:4]]
    assert wi[:4] == known_i, [x.hex() for x in wi[:4]]
    # and libm at the same double argument agrees wherever it is correctly rounded (it is, for these)
    for k in range(63):
        assert wr[k] == math.cos(2.0 * math.pi / 2.0 ** k) and wi[k] == -math.sin(2.0 * math.pi / 2.0 ** k), k
    for name, tab in (("CM3_W_SUB_N_R", wr), ("CM3_W_SUB_N_I", wi)):
        print("static const double %s[63] = {" % name)
        for i in range(0, 63, 3):
            print("    " + ", ".join(x.hex() for x in tab[i:i + 3]) + ",")
        print("};")


if __name__ == "__main__":
    main()
