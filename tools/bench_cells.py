#!/usr/bin/env python3
"""Every (size, format, hop, window) cell of the packed family's dispatch at a glance: lines/s and the fraction of 8 TB/s --
to find the cells that have no register-reuse variant and re-read their overlap through L2.   usage: bench_cells.py [sizes...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa

st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)

def timeit(fn, reps=8, warm=5):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

sizes = [int(x) for x in sys.argv[1:]] or [512, 1024, 4096]
for nfft in sizes:
    for dt in ("cf32_le", "ci16_le", "cu8", "cf32_be", "ci16_be"):
        row = []
        for win in (0, 1):
            for hop in (nfft, nfft // 2, nfft // 4, nfft - 7):
                bps = sa.bytes_per_sample(dt)
                n = min(((1 << 28) - nfft) // hop + 1, (1 << 30) // (nfft * 4))
                iq = svc.synth_iq(dt, 7, 0, (n - 1) * hop + nfft)
                out = torch.empty((n, nfft), dtype=torch.float32, device="cuda")
                ms = timeit(lambda: svc.compute_waterfall(iq, 0, nfft, dt, n, hop=hop, window=win, out=out))
                row.append("%.3f (%5.0fM)" % (n * (hop * bps + nfft * 4) / ms / 1e6 / 8000, n / ms / 1e3))
                del iq, out; torch.cuda.empty_cache()
        print("%5d %-8s rect: N %s  N/2 %s  N/4 %s  N-7 %s | hann: N %s  N/2 %s  N/4 %s  N-7 %s" % (nfft, dt, *row), flush=True)
