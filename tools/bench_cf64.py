#!/usr/bin/env python3
"""The fp64 (strict-parity) pipeline: cf64 recordings and fp64 outputs at 1024 / 4096 / 8192 / 16384 points,
50 % overlap (development tool; HIP-event medians).   python tools/bench_cf64.py [log2_samples=27] [nfft ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa

log2s = int(sys.argv[1]) if len(sys.argv) > 1 else 27
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

for nfft in ([int(x) for x in sys.argv[2:]] or (1024, 4096, 8192, 16384)):
    for dt, fmt, label in (("cf64_le", sa.OUT_DB20_F64, "cf64->f64"), ("cf64_le", sa.OUT_DB20_F32, "cf64->f32"),
                           ("cf32_le", sa.OUT_DB20_F64, "cf32->f64")):
        hop, S, bps = nfft // 2, 1 << log2s, sa.bytes_per_sample(dt)
        n = (S - nfft) // hop + 1
        iq = svc.synth_iq(dt, 7, 0, S)
        out = torch.empty((n, nfft), dtype=torch.float32 if fmt < 2 else torch.float64, device="cuda")
        ms = timeit(lambda: svc.compute_waterfall(iq, 0, nfft, dt, n, hop=hop, out_fmt=fmt, out=out))
        b = n * (hop * bps + nfft * (4 if fmt < 2 else 8))
        print("%-10s n=%-5d %9d lines  %8.3f ms  %8.2f Mlines/s  %7.0f GB/s algorithmic (%.1f%% of 8 TB/s)"
              % (label, nfft, n, ms, n / ms / 1e3, b / ms / 1e6, b / ms / 1e6 / 80), flush=True)
        del iq, out; torch.cuda.empty_cache()
