#!/usr/bin/env python3
"""Writes the golden fixtures tests/golden/*.npz.

The reference (Java, third-party FFT jar, no JDK in the build image) cannot be
run to produce vectors, and it ships none.  These fixtures are therefore made
from the CPU oracle (oracle/spec_oracle.c) AFTER checking it against an
independent FFT (numpy.fft) on the very same input; each file holds the input
bytes, the parameters and the expected dB lines (float64).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import spec_oracle as so  # noqa: E402

CASES = []
for seed, dt in enumerate(["cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8", "cf64_le", "cf64_be"], start=1):
    for nfft in (64, 1024, 4096):
        CASES.append((dt, nfft, nfft // 2, so.WIN_RECT, 4, seed))
CASES.append(("cf32_le", 1024, 1024, so.WIN_RECT, 5, 9))      # the reference's own hop == nfft
CASES.append(("cf32_le", 4096, 1024, so.WIN_HANN, 6, 10))     # 75 % overlap, Hann
CASES.append(("ci16_le", 16384, 8192, so.WIN_RECT, 3, 11))

for dt, nfft, hop, window, n_lines, seed in CASES:
    n = (n_lines - 1) * hop + nfft
    iq = so.synth_iq(dt, seed, 1234, n)
    db = so.waterfall(iq, 0, dt, nfft, hop, n_lines + 1, window)   # + one EOF line (-150)
    chk = so.np_waterfall(iq, 0, dt, nfft, hop, n_lines + 1, window)
    lin, lin2 = 10 ** (db / 20), 10 ** (chk / 20)
    assert np.abs(lin - lin2).max() <= 1e-11 * lin2.max(), (dt, nfft)
    name = "wf_%s_n%d_h%d_w%d.npz" % (dt, nfft, hop, window)
    np.savez_compressed(os.path.join(HERE, name), iq=iq, db=db, datatype=dt, nfft=nfft, hop=hop,
                        window=window, seed=seed)
    print("wrote", name, iq.nbytes, "B in,", db.nbytes, "B out")
