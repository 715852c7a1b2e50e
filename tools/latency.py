"""Host-buffer call latencies of the drop-in entry points (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spectral_analyzer_amd as sa
from oracle import spec_oracle as so
svc = sa.SpectralService(0)
for dt, nfft in (("ci16_le", 1024), ("cf32_le", 4096), ("cf32_le", 65536)):
    iq = so.synth_iq(dt, 1, 0, 1000 * nfft if nfft <= 4096 else 4 * nfft)
    for _ in range(20): svc.compute_magnitudes(iq, 0, nfft, dt)
    t0 = time.perf_counter(); n = 200
    for i in range(n): svc.compute_magnitudes(iq, 0, nfft, dt)
    one = (time.perf_counter() - t0) / n
    W = 1000 if nfft <= 4096 else 4
    for _ in range(3): svc.compute_waterfall(iq, 0, nfft, dt, W)
    t0 = time.perf_counter()
    for i in range(10): svc.compute_waterfall(iq, 0, nfft, dt, W)
    wf = (time.perf_counter() - t0) / 10
    for _ in range(3): svc.waterfall_render(iq, 0, nfft, dt, W, 600, 1e6)
    t0 = time.perf_counter()
    for i in range(10): svc.waterfall_render(iq, 0, nfft, dt, W, 600, 1e6)
    wr = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for i in range(3): so.waterfall(iq, 0, dt, nfft, nfft, W)
    cpu = (time.perf_counter() - t0) / 3
    # the UNMODIFIED reference loop (MC:982-993): one computeMagnitudes call per slice, slice after slice
    bps = so.bytes_per_sample(dt)
    walk = {}
    for ra in (0, 256):
        svc.set_option("readahead_lines", ra)
        for _ in range(2): [svc.compute_magnitudes(iq, t * nfft * bps, nfft, dt) for t in range(W)]
        t0 = time.perf_counter()
        for _ in range(3): [svc.compute_magnitudes(iq, t * nfft * bps, nfft, dt) for t in range(W)]
        walk[ra] = (time.perf_counter() - t0) / 3
    print("%s nfft %5d: computeMagnitudes %.1f us/call | %d-line redraw: batched %.2f ms, batched+render %.2f ms, "
          "%d per-slice calls %.1f ms without / %.2f ms with read-ahead, CPU oracle 1 thread %.1f ms"
          % (dt, nfft, one * 1e6, W, wf * 1e3, wr * 1e3, W, walk[0] * 1e3, walk[256] * 1e3, cpu * 1e3))
