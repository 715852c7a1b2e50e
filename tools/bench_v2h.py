#!/usr/bin/env python3
"""32768-point fp32 lines: the single-workgroup kernel (spec_k_v2h.hip, default) against the four-step team kernel
("large_single" = 0).  Development tool; prints one line per case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa

st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)

def timeit(fn, reps=8, warm=6):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

def spectro(dt, nfft, hop, log2s, window=0, single=1, lpw=0):
    bps = sa.bytes_per_sample(dt); S = 1 << log2s
    n = (S - nfft) // hop + 1
    iq = svc.synth_iq(dt, 7, 0, S)
    out = torch.empty((n, nfft), dtype=torch.float32, device="cuda")
    svc.set_option("large_single", single); svc.set_option("lines_per_wg", lpw)
    ms = timeit(lambda: svc.compute_waterfall(iq, 0, nfft, dt, n, hop=hop, window=window, out=out))
    svc.set_option("large_single", 1); svc.set_option("lines_per_wg", 0)
    b = n * (hop * bps + nfft * 4)
    print("%-9s n=%d hop=%-6d 2^%d win=%d single=%d lpw=%-2d %8d lines %8.3f ms %8.2f Mlines/s %6.0f GB/s (%.1f%% of 8 TB/s)" % (
        dt, nfft, hop, log2s, window, single, lpw, n, ms, n / ms / 1e3, b / ms / 1e6, b / ms / 1e6 / 80), flush=True)
    del iq, out; torch.cuda.empty_cache()

which = sys.argv[1:] or ["base"]
if "base" in which:
    for single in (1, 0):
        spectro("cf32_le", 32768, 16384, 28, single=single)
        spectro("cf32_le", 32768, 16384, 30, single=single)
        spectro("ci16_le", 32768, 16384, 30, single=single)
    spectro("cf32_le", 32768, 32768, 30)
    spectro("cf32_le", 32768, 16384, 30, window=1)
    spectro("cu8", 32768, 16384, 30)
if "pf" in which:
    spectro("cf32_le", 32768, 16384, 30)
    spectro("cf32_le", 32768, 32768, 30)
if "lpw" in which:
    for lpw in (1, 2, 4, 8, 16, 32):
        spectro("cf32_le", 32768, 16384, 30, lpw=lpw)
