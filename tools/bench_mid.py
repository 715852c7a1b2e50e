#!/usr/bin/env python3
"""16384-point (or, `python tools/bench_mid.py 8192`, 8192-point) fp32 lines: the family's kernel (register reuse of the
overlap) against the half-line kernel of spec_k_v2h.hip at that size ("mid_single" / "small_single" = 1: 256-thread
workgroups, two / three per CU).  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
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
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
KNOB = "mid_single" if NF == 16384 else "small_single"
for dt in ("cf32_le", "ci16_le", "cu8", "cf32_be"):
    for hop in (NF // 4, NF // 2, NF * 3 // 4 + 57, NF):
        for win in (0, 1):
            S = 1 << 29
            n = (S - NF) // hop + 1
            iq = svc.synth_iq(dt, 7, 0, S)
            out = torch.empty((n, NF), dtype=torch.float32, device="cuda")
            bps = sa.bytes_per_sample(dt)
            r = []
            for mid in (0, 1):
                svc.set_option(KNOB, mid)
                ms = timeit(lambda: svc.compute_waterfall(iq, 0, NF, dt, n, hop=hop, window=win, out=out))
                r.append(n * (hop * bps + NF * 4) / ms / 1e6 / 80)
            print("%-8s hop %5d win %d: family %.1f %%  half-line %.1f %%  %s" % (dt, hop, win, r[0], r[1], "<-- half-line" if r[1] > r[0] * 1.01 else ""), flush=True)
            del iq, out; torch.cuda.empty_cache()
