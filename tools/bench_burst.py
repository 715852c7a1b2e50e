#!/usr/bin/env python3
"""Timings of the burst chain on one GPU, device-resident, beside the
C oracle on the host (one thread: the reference runs these loops on the JavaFX thread).

    python tools/bench_burst.py [log2_samples=24]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
from oracle import spec_oracle as so

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log2n
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
edc = sa.ExtractDownConvertService(svc)

def timeit(fn, reps=10, warm=4):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

def cpu(fn):
    t0 = time.perf_counter(); fn(); return (time.perf_counter() - t0) * 1e3

dt, down, f_off, alpha = "ci16_le", 8, 0.0731, 0.1
iq = svc.synth_iq(dt, 11, 0, n)
host = iq[: (1 << min(log2n, 22)) * 4].cpu().numpy()   # CPU sample: at most 2^22 samples
nc = len(host) // 4
rows = []
ms = timeit(lambda: edc.extract_iq(iq, 0, n, dt))
rows.append(("reader (EDC:60-97) ci16 -> 2 x f64", ms, n * (4 + 16), cpu(lambda: so.extract_iq(host, 0, nc, dt)), nc))
re_c, im_c = so.extract_iq(host, 0, nc, dt)
for fast in (True, False):
    ms = timeit(lambda: edc.extract_and_down_convert(iq, 0, n, dt, f_off, down, fast))
    rows.append(("extractAndDownConvert down=8 %s" % ("fast" if fast else "lpf"), ms, n * 4 + n // down * 16,
                 cpu(lambda: so.down_convert(re_c, im_c, f_off, down, 0 if fast else 1)), nc))
d = edc.extract_and_down_convert(iq, 0, n, dt, f_off, 1, True)   # full-rate burst for the traces
dc = [x.cpu().numpy()[:nc] for x in d]
ms = timeit(lambda: svc.magnitude_trace(d, alpha))
rows.append(("magnitude trace (ADC:219-246)", ms, n * (16 + 8), cpu(lambda: so.magnitude_trace(dc[0], dc[1], alpha)), nc))
ms = timeit(lambda: svc.inst_freq_trace(d, alpha, 1e6, 0.0))
rows.append(("frequency trace (ADC:256-284)", ms, n * (16 + 8), cpu(lambda: so.inst_freq_trace(dc[0], dc[1], alpha, 1e6, 0.0)), nc))
print("burst of 2^%d %s samples; CPU column: C oracle, one thread, 2^%d samples" % (log2n, dt, int(np.log2(nc))))
for name, ms, b, cms, cn in rows:
    print("%-44s %8.3f ms  %8.1f Msamples/s  %6.0f GB/s algorithmic (%4.1f%% of 8 TB/s) | CPU %7.1f Msamples/s  x%.0f"
          % (name, ms, n / ms / 1e3, b / ms / 1e6, b / ms / 1e6 / 80, cn / cms / 1e3, (n / ms) / (cn / cms)))
