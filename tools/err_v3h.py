#!/usr/bin/env python3
"""Whose rounding error is it?  One 16384-point line (cf32 and cf64 samples) transformed by (a) the C oracle (commons-math3's
radix-2 transform restated), (b) the single-workgroup fp64 kernel, (c) the four-step path it replaced, each compared with a DFT in
long double (numpy, O(N^2), a minute of CPU).  Errors relative to the line's peak magnitude M.  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spectral_analyzer_amd as sa
from oracle import spec_oracle as so
so.build()
svc = sa.SpectralService(0)
N = 16384
for dt in ("cf32_le", "cf64_le"):
    for seed in (5, 6):
        iq = so.synth_iq(dt, seed, 0, N)
        x = (iq.view(np.float32) if dt == "cf32_le" else iq.view(np.float64)).astype(np.longdouble).reshape(-1, 2)
        xc = x[:, 0] + 1j * x[:, 1]
        n = np.arange(N, dtype=np.longdouble)
        X = np.empty(N, dtype=np.clongdouble)
        two_pi = 2 * np.pi.__class__(np.longdouble(3.14159265358979323846264338327950288))
        for k0 in range(0, N, 256):
            k = np.arange(k0, k0 + 256, dtype=np.longdouble)[:, None]
            ph = (k * n[None, :]) % N
            X[k0:k0 + 256] = (np.exp(-1j * (two_pi * ph / N)) * xc[None, :]).sum(axis=1)
        P = np.fft.fftshift((X.real ** 2 + X.imag ** 2)).astype(np.float64)   # shifted power, reference
        M2 = P.max()
        res = {"oracle": so.waterfall(iq, 0, dt, N, N, 1, 0, power=True)[0]}
        for name, single in (("single workgroup", 1), ("four-step", 0)):
            svc.set_option("large_single", single)
            res[name] = svc.compute_waterfall(iq, 0, N, dt, 1, hop=N, out_fmt=sa.OUT_POW_F64)[0]
        svc.set_option("large_single", 1)
        for name, p in res.items():
            mag_err = np.abs(np.sqrt(p) - np.sqrt(P)) / np.sqrt(M2)
            weak = np.sqrt(P) >= 1e-5 * np.sqrt(M2)
            db_err = np.abs(10 * np.log10(p[weak]) - 10 * np.log10(P[weak]))
            print("%-8s seed %d  %-17s max |d|X||/M = %.2e   max dB error on bins >= 1e-5 M = %.2e" % (dt, seed, name, mag_err.max(), db_err.max()), flush=True)
