#!/usr/bin/env python3
"""One number per call: the 65536-point pair kernel (or, with 15 as the first argument, the 32768-point fp64 one) on the bench's
n65536f shape, for A/B runs of variant libraries (SPEC_LIB_VARIANT=<name>).  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa

st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
cases = sys.argv[1:] or ["cf32_le:32768:0", "ci16_le:32768:0", "cf32_le:32768:1", "cf32_le:65536:0"]
NFFT = 65536
for case in cases:
    dt, hop, win = case.split(":"); hop = int(hop); win = int(win)
    S = 1 << (30 if sa.bytes_per_sample(dt) == 8 else 30)
    n = (S - NFFT) // hop + 1
    iq = svc.synth_iq(dt, 7, 0, S)
    out = torch.empty((n, NFFT), dtype=torch.float32, device="cuda")
    fn = lambda: svc.compute_waterfall(iq, 0, NFFT, dt, n, hop=hop, window=win, out=out)
    for _ in range(5): fn()
    ev = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    by = n * (hop * sa.bytes_per_sample(dt) + NFFT * 4)
    print("%-12s %-22s %7.3f ms  %.3f of 8 TB/s" % (os.environ.get("SPEC_LIB_VARIANT", "product"), case, ms, by / ms / 8e9), flush=True)
    del iq, out; torch.cuda.empty_cache()
