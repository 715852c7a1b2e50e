#!/usr/bin/env python3
"""spec_waterfall_render on a device-resident recording: fused (compact tile) against the two-pass form.
    python tools/bench_render.py [log2_samples=28]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa

log2s = int(sys.argv[1]) if len(sys.argv) > 1 else 28
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)

def timeit(fn, reps=8, warm=4):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

for dt, nfft, hop, height, S in (("cf32_le", 4096, 4096, 600, 1 << log2s), ("cf32_le", 4096, 2048, 1024, 1 << log2s),
                                 ("ci16_le", 1024, 1024, 512, 1 << log2s), ("cf32_le", 4096, 4096, 600, 1000 * 4096)):
    iq = svc.synth_iq(dt, 7, 0, S)
    width = (S - nfft) // hop + 1
    res = {}
    for fused in (1, 0):
        svc.set_option("render_fused", fused)
        res[fused] = timeit(lambda: svc.waterfall_render(iq, 0, nfft, dt, width, height, 1e6, hop=hop))
    svc.set_option("render_fused", 1)
    bps = sa.bytes_per_sample(dt)
    print("%-8s nfft %5d hop %5d height %4d width %7d: fused %8.3f ms (%6.0f GB/s of input)  two-pass %8.3f ms  x%.2f"
          % (dt, nfft, hop, height, width, res[1], width * hop * bps / res[1] / 1e6, res[0], res[0] / res[1]), flush=True)
    del iq; torch.cuda.empty_cache()
