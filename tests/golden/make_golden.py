#!/usr/bin/env python3
"""Writes the golden fixtures tests/golden/*.npz.

The reference (Java, third-party FFT jar, no JDK in the build image) cannot be
run to produce vectors, and it ships none.  These fixtures are therefore made
from the CPU oracle (oracle/spec_oracle.c) -- since round 3 with commons-math3
3.6.1's transform restated as published (twiddles by recurrence, Complex.abs()),
i.e. the reference's own rounding behaviour -- AFTER checking it against an
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

# ---- burst chain: reader, down-converter, traces ------------------
# Checked against straight numpy restatements of the cited Java loops before being written.
def np_ema(x, alpha):
    v = np.empty_like(x)
    for i in range(len(x)):
        v[i] = x[i] if i == 0 else alpha * x[i] + (1 - alpha) * v[i - 1]
    return v


for dt, seed in (("ci16_le", 21), ("cf32_le", 22), ("cu8", 23), ("cf64_be", 24)):
    n, start, count, down, f_off, alpha, fs, fc = 6000, 37, 5000, 8, 0.0731, 0.15, 2.4e6, 915e6
    iq = so.synth_iq(dt, seed, 99, n)
    re, im = so.extract_iq(iq, start, count, dt)
    z = so.np_decode(iq, start * so.bytes_per_sample(dt), count, dt)
    assert np.array_equal(re, z.real) and np.array_equal(im, z.imag), dt
    out = {}
    for mode in (0, 1):
        dr, di = so.down_convert(re, im, f_off, down, mode)
        h, c = so.down_convert_taps(down, mode)
        t = f_off * np.arange(count)
        xm = z * np.exp(-2j * np.pi * (t - np.floor(t)))
        pad = np.concatenate([np.zeros(len(h), complex), xm, np.zeros(len(h), complex)])
        chk = np.array([sum(h[k] * pad[len(h) + m * down + c - k] for k in range(len(h))) for m in range(count // down)])
        assert np.abs((dr + 1j * di) - chk).max() <= 1e-13, (dt, mode)
        out["dc%d_re" % mode], out["dc%d_im" % mode] = dr, di
    mag = so.magnitude_trace(out["dc0_re"], out["dc0_im"], alpha)
    assert np.abs(mag - 20 * np.log10(np_ema(np.hypot(out["dc0_re"], out["dc0_im"]), alpha))).max() <= 1e-11
    frq = so.inst_freq_trace(out["dc0_re"], out["dc0_im"], alpha, fs / down, fc)
    ph = np.arctan2(out["dc0_im"], out["dc0_re"])
    d = ph[1:] - ph[:-1]
    d = np.where(d > np.pi, d - 2 * np.pi, np.where(d < -np.pi, d + 2 * np.pi, d))
    assert np.abs(frq - (np_ema(d / (2 * np.pi) * (fs / down), alpha) + fc)).max() <= 1e-6
    name = "burst_%s.npz" % dt
    np.savez_compressed(os.path.join(HERE, name), iq=iq, datatype=dt, start=start, count=count, down=down,
                        freq_off=f_off, alpha=alpha, fs=fs / down, center=fc, re=re, im=im, mag=mag, freq=frq, **out)
    print("wrote", name)
