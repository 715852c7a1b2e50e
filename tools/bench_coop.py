#!/usr/bin/env python3
"""The wave-cooperative kernel (spec_k_v2n.hip) against the packed family's own kernel at the sizes both can take:
"coop_256" = 0 (family) / 1 (cooperative kernel), every format, two hops, one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa

st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)

def timeit(fn, reps=10, warm=6):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

nfft = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lg = nfft.bit_length() - 1
for dt in ("cf32_le", "ci16_le", "cu8", "ci8", "cf32_be", "ci16_be"):
    for hop, win in ((nfft, 0), (nfft // 2, 0), (nfft // 4, 0), (nfft // 2, 1), (nfft - 7, 0)):
        bps = sa.bytes_per_sample(dt); S = 1 << 29
        n = (S - nfft) // hop + 1
        n = min(n, 1 << 22)
        iq = svc.synth_iq(dt, 7, 0, (n - 1) * hop + nfft)
        out = torch.empty((n, nfft), dtype=torch.float32, device="cuda")
        res, outs = [], []
        for knob in (0, 1):
            svc.set_option("coop_256", knob)
            ms = timeit(lambda: svc.compute_waterfall(iq, 0, nfft, dt, n, hop=hop, window=win, out=out))
            res.append(n * (hop * bps + nfft * 4) / ms / 1e6 / 8000)
            torch.cuda.synchronize()
            outs.append(out[:4096].clone())
        big = outs[0] > outs[0].max() - 60.0  # bins within 60 dB of the peak
        diff = float((outs[0] - outs[1]).abs()[big].max())
        print("%-8s n=%d hop=%-4d win=%d  %8d lines   family %.3f   cooperative %.3f   x%.2f   max |dB diff| (bins within 60 dB) %.2e" % (
            dt, nfft, hop, win, n, res[0], res[1], res[1] / res[0], diff), flush=True)
        del iq, out; torch.cuda.empty_cache()
