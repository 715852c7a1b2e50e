#!/usr/bin/env python3
"""Development aid: where the single-workgroup fp64 kernel differs from the oracle (line, bin, size of the error)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
from oracle import spec_oracle as oracle
oracle.build()
svc = sa.SpectralService(0)
NFFT = 16384
for datatype, hop, window, fmt in (("cf64_be", 8192, 1, sa.OUT_POW_F64), ("cf64_be", 8192, 1, sa.OUT_POW_F32), ("cf64_be", 8192, 1, sa.OUT_DB20_F64), ("cf64_be", 16384, 1, sa.OUT_POW_F64), ("cf64_be", 5000, 1, sa.OUT_POW_F64)):
    n_lines = 9
    iq = oracle.synth_iq(datatype, seed=hop + window + 2, first_sample=1, n_samples=(n_lines - 1) * hop + NFFT)
    ref = oracle.waterfall(iq, 0, datatype, NFFT, hop, n_lines, window=window, power=fmt in (sa.OUT_POW_F64, sa.OUT_POW_F32))
    d = torch.from_numpy(iq).cuda()
    for lpw in (4, 2):
        svc.set_option("lines_per_wg", lpw)
        outs = [svc.compute_waterfall(d, 0, NFFT, datatype, n_lines, hop=hop, window=window, out_fmt=fmt).cpu().numpy().astype(np.float64) for _ in range(3)]
        got = outs[0]
        print(datatype, hop, window, fmt, "lpw", lpw, "repeatable:", all(np.array_equal(outs[0], o, equal_nan=True) for o in outs[1:]))
        err = np.abs(got - ref) / np.abs(ref).max(axis=1, keepdims=True)
        thr = 1e-5 if fmt in (sa.OUT_POW_F32,) else 1e-9
        for ln in range(n_lines):
            cols = np.nonzero(err[ln] > thr)[0]
            if len(cols) == 0: continue
            k = (cols // 2 - NFFT // 4) % (NFFT // 2)
            print("  line", ln, "bad", len(cols), "parity", set(cols % 2), "t", sorted(set(k % 512))[:40], "m", sorted(set(k // 512)))
            c = cols[0]
            print("    bin", c, "got", got[ln, c], "ref", ref[ln, c], "| same value elsewhere:", np.argwhere(got == got[ln, c])[:4].tolist())
