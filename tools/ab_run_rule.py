#!/usr/bin/env python3
"""A/B of run lengths by alternating measurements on one box: `ab_run_rule.py hopn` -- runs of 32 against the automatic rule at
hop = nfft (1024 / 2048 / 4096 points, three recording sizes); `ab_run_rule.py half K...` -- runs of 32 against runs of K at 50 %
overlap (256 ... 4096 points).  Fractions of 8 TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
def timeit(fn, reps=12, warm=8):
    for _ in range(warm): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
F64 = os.environ.get("AB_F64") == "1"   # the fp64 pipeline (cf64 / ci16 in, DB20_F64 out) instead of the fp32 one
mode = sys.argv[1] if len(sys.argv) > 1 else "hopn"
ks = [int(x) for x in sys.argv[2:]] or [0]
for dt in (("cf64_le", "ci16_le") if F64 else ("cf32_le", "ci16_le")):
  for nfft in ([int(x) for x in os.environ["AB_SIZES"].split(",")] if os.environ.get("AB_SIZES") else (1024, 2048, 4096) if mode == "hopn" else (256, 512, 1024, 2048, 4096)):
    for lg in ((28,) if F64 else (26, 28, 30) if mode == "hopn" else (28, 30)):
        S = 1 << lg; hop = nfft if mode == "hopn" else nfft // 2; n = (S - nfft) // hop + 1; bps = sa.bytes_per_sample(dt)
        iq = svc.synth_iq(dt, 7, 0, S); out = torch.empty((n, nfft), dtype=torch.float64 if F64 else torch.float32, device="cuda")
        res = {}
        for rep in range(2):
            for k in [32] + ks:
                svc.set_option("lines_per_wg", k)
                ms = timeit(lambda: svc.compute_waterfall(iq, 0, nfft, dt, n, hop=hop, out=out, out_fmt=sa.OUT_DB20_F64 if F64 else sa.OUT_DB20_F32))
                res.setdefault(k, []).append(n * (hop * bps + nfft * (8 if F64 else 4)) / ms / 1e6 / 8000)
        svc.set_option("lines_per_wg", 0)
        print("%-8s n=%-5d hop=%-5d 2^%d samples: " % (dt, nfft, hop, lg) + "   ".join("%s: %.3f %.3f" % ("rule" if k == 0 else "runs of %d" % k, *res[k]) for k in [32] + ks), flush=True)
        del iq, out; torch.cuda.empty_cache()
