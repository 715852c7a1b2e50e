#!/usr/bin/env python3
"""16384-point fp64 lines (the strict-parity pipeline): the single-workgroup kernel of spec_k_v3h.hip against the four-step
team kernel it replaces in the default dispatch ("large_single" = 0), formats x hops x window x output width.
Development tool; HIP-event medians; algorithmic bytes = new samples + the line's bins.   python tools/bench_v3h.py [log2_samples=28]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
log2s = int(sys.argv[1]) if len(sys.argv) > 1 else 28
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
N = 16384
def timeit(fn, reps=8, warm=5):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for dt, fmt in (("cf64_le", sa.OUT_DB20_F64), ("cf64_le", sa.OUT_DB20_F32), ("cf64_be", sa.OUT_DB20_F64), ("cf32_le", sa.OUT_DB20_F64),
                ("ci16_le", sa.OUT_DB20_F64), ("cu8", sa.OUT_POW_F64)):
    bps = sa.bytes_per_sample(dt)
    S = (1 << log2s) * 16 // max(bps, 4) // 4          # about the same bytes of recording for every format
    for hop in (8192, 16384, 4096, 12345):
        for win in (0, 1):
            if win and hop not in (8192, 16384): continue
            n = (S - N) // hop + 1
            iq = svc.synth_iq(dt, 7, 0, S)
            esz = 8 if fmt >= sa.OUT_DB20_F64 else 4
            out = torch.empty((n, N), dtype=torch.float64 if esz == 8 else torch.float32, device="cuda")
            r = []
            for single in (0, 1):
                svc.set_option("large_single", single)
                ms = timeit(lambda: svc.compute_waterfall(iq, 0, N, dt, n, hop=hop, window=win, out_fmt=fmt, out=out))
                r.append(n * (min(hop, N) * bps + N * esz) / ms / 1e6 / 80)
            svc.set_option("large_single", 1)
            print("%-8s -> f%d hop %5d win %d %7d lines: team %.1f %%  single workgroup %.1f %% of 8 TB/s  %s"
                  % (dt, esz * 8, hop, win, n, r[0], r[1], "" if r[1] > r[0] else "<-- team faster"), flush=True)
            del iq, out; torch.cuda.empty_cache()
