#!/usr/bin/env python3
"""32768-point fp64 lines: the paired kernel (spec_k_v3h.hip v3q_kernel, "large_pair" = 1) against the four-step team kernel
("large_pair" = 0), per format / output / hop / window.  Development tool; prints one line per case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
from spectral_analyzer_amd import _lib as L

st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
NFFT = 32768

def timeit(fn, reps=8, warm=5):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

def spectro(dt, hop, log2s, window=0, f64=True):
    bps = sa.bytes_per_sample(dt); S = 1 << log2s
    n = (S - NFFT) // hop + 1
    iq = svc.synth_iq(dt, 7, 0, S)
    out = torch.empty((n, NFFT), dtype=torch.float64 if f64 else torch.float32, device="cuda")
    fmt = L.OUT_DB20_F64 if f64 else L.OUT_DB20_F32
    res = []
    for pair in (0, 1):
        svc.set_option("large_pair", pair)
        res.append(timeit(lambda: svc.compute_waterfall(iq, 0, NFFT, dt, n, hop=hop, window=window, out_fmt=fmt, out=out)))
    svc.set_option("large_pair", 1)
    b = n * (hop * bps + NFFT * (8 if f64 else 4))
    print("%-8s -> %s hop=%-6d 2^%d win=%d %7d lines  team %7.3f ms (%.3f)  pair %7.3f ms (%.3f of 8 TB/s)  x%.2f" % (
        dt, "f64" if f64 else "f32", hop, log2s, window, n, res[0], b / res[0] / 8e9, res[1], b / res[1] / 8e9, res[0] / res[1]), flush=True)
    del iq, out; torch.cuda.empty_cache()

for dt, log2s in (("cf64_le", 28), ("cf64_be", 28), ("cf32_le", 29), ("ci16_le", 29), ("cu8", 29)):
    for hop, win in ((16384, 0), (32768, 0), (16384, 1), (20000, 0)):
        spectro(dt, hop, log2s, window=win)
spectro("cf64_le", 16384, 28, f64=False)
