#!/usr/bin/env python3
"""Lines per sub-line run ("lines_per_wg") against the automatic choice (0), size by size, 2^30-sample recordings at 50 % overlap
and at the reference's hop = nfft: fraction of 8 TB/s.   usage: bench_run_len.py [sizes...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa

st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)

def timeit(fn, reps=10, warm=8):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

args = [a for a in sys.argv[1:] if not a.startswith("--")]
sizes = [int(x) for x in args] or [256, 512, 1024, 2048, 4096]
RUNS = (0, 2, 4, 8, 16) if "--short" in sys.argv else (0, 4, 8, 12, 16, 24, 32)
HOPS = (1,) if "--hopn" in sys.argv else (2, 1)
print("%-26s" % "lines_per_wg" + "".join("%8d" % r for r in RUNS))
for dt in ("cf32_le", "ci16_le"):
    for nfft in sizes:
        for hop in [nfft // d for d in HOPS]:
            bps = sa.bytes_per_sample(dt); S = 1 << 30
            n = (S - nfft) // hop + 1
            iq = svc.synth_iq(dt, 7, 0, S)
            out = torch.empty((n, nfft), dtype=torch.float32, device="cuda")
            row = []
            for r in RUNS:
                svc.set_option("lines_per_wg", r)
                ms = timeit(lambda: svc.compute_waterfall(iq, 0, nfft, dt, n, hop=hop, out=out))
                row.append(n * (hop * bps + nfft * 4) / ms / 1e6 / 8000)
            svc.set_option("lines_per_wg", 0)
            print("%-8s n=%-5d hop=%-5d " % (dt, nfft, hop) + "".join("%8.3f" % x for x in row), flush=True)
            del iq, out; torch.cuda.empty_cache()
