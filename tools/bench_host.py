#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) rate of spec_waterfall, as the JNI path drives it: input in host
memory (the mapped file), output to a host array.  python tools/bench_host.py [log2_samples=27]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa

log2s = int(sys.argv[1]) if len(sys.argv) > 1 else 27
svc = sa.SpectralService(0)
for dt, nfft, hop in (("cf32_le", 4096, 2048), ("ci16_le", 4096, 2048), ("cf32_le", 1024, 1024)):
    S = 1 << log2s
    bps = sa.bytes_per_sample(dt)
    host = svc.synth_iq(dt, 3, 0, S).cpu().numpy()
    n = (S - nfft) // hop + 1
    out = np.empty((n, nfft), dtype=np.float32)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        svc.compute_waterfall(host, 0, nfft, dt, n, hop=hop, out=out)
        best = min(best, time.perf_counter() - t0)
    moved = host.nbytes + out.nbytes
    print("%-8s nfft %5d hop %5d  %8d lines  %8.1f ms  %6.2f Mlines/s  %5.1f GB/s over PCIe (in %.2f GB + out %.2f GB)"
          % (dt, nfft, hop, n, best * 1e3, n / best / 1e6, moved / best / 1e9, host.nbytes / 1e9, out.nbytes / 1e9), flush=True)
