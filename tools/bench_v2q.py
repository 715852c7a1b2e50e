#!/usr/bin/env python3
"""65536-point fp32 lines: the paired single-workgroup kernels (spec_k_v2q.hip, "large_pair" = 1) against the four-step team
kernel ("large_pair" = 0), per format / hop / window.  Development tool; prints one line per case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa

st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
NFFT = 65536

def timeit(fn, reps=8, warm=5):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

def spectro(dt, hop, log2s, window=0, **opts):
    bps = sa.bytes_per_sample(dt); S = 1 << log2s
    n = (S - NFFT) // hop + 1
    iq = svc.synth_iq(dt, 7, 0, S)
    out = torch.empty((n, NFFT), dtype=torch.float32, device="cuda")
    res = []
    for pair in (0, 1):
        svc.set_option("large_pair", pair)
        for k, v in opts.items(): svc.set_option(k, v)
        ms = timeit(lambda: svc.compute_waterfall(iq, 0, NFFT, dt, n, hop=hop, window=window, out=out))
        res.append(ms)
    svc.set_option("large_pair", 1)
    b = n * (hop * bps + NFFT * 4)
    print("%-9s hop=%-6d 2^%d win=%d %s %7d lines  team %7.3f ms (%.3f)  pair %7.3f ms (%.3f of 8 TB/s)  x%.2f" % (
        dt, hop, log2s, window, opts or "", n, res[0], b / res[0] / 8e9, res[1], b / res[1] / 8e9, res[0] / res[1]), flush=True)
    del iq, out; torch.cuda.empty_cache()

which = sys.argv[1:] or ["base"]
if "base" in which:
    for dt in ("cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8"):
        for hop, win in ((32768, 0), (65536, 0), (32768, 1), (16384, 0), (40000, 0)):
            spectro(dt, hop, 29 if sa.bytes_per_sample(dt) == 8 else 30, window=win)
if "short" in which:
    for log2s in (24, 26, 28):
        spectro("cf32_le", 32768, log2s)
        spectro("ci16_le", 32768, log2s)
if "ilv" in which:
    for ilv in (0, 1):
        spectro("cf32_le", 32768, 30, pair_interleave=ilv)
        spectro("ci16_le", 32768, 30, pair_interleave=ilv)
