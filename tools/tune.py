#!/usr/bin/env python3
"""In-process A/B of run lengths (and generic vs packed kernels) on the cfg2 / cfg3 workloads:
interleaved rounds at a settled clock, median/min.  Development tool, not part of the product.

    python tools/tune.py --workload cfg2 --lpw 0,16,64 --rounds 5 [--generic] [--window 1] [--hop 1024]
"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spectral_analyzer_amd as sa
from spectral_analyzer_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--lpw", default="0")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--log2", type=int, default=30)
ap.add_argument("--generic", action="store_true", help="also time the generic kernel")
ap.add_argument("--window", type=int, default=0)
ap.add_argument("--hop", type=int, default=2048)
args = ap.parse_args()
dt = {"cfg2": "cf32_le", "cfg3": "ci16_le"}[args.workload]
nfft, hop = 4096, args.hop
bps = sa.bytes_per_sample(dt)
S = 1 << args.log2
n_lines = (S - nfft) // hop + 1
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
svc = sa.SpectralService(0, stream=st.cuda_stream)
iq = svc.synth_iq(dt, 0x5EC7A11A, 0, S)
out = torch.empty((n_lines, nfft), dtype=torch.float32, device="cuda")
ref = torch.empty_like(out)
svc.set_option("force_generic", 1)
svc.compute_waterfall(iq, 0, nfft, dt, n_lines, hop=hop, window=args.window, out=ref)
torch.cuda.synchronize()
cfgs = [("generic", None, 0)] if args.generic else []
for l in [int(x) for x in args.lpw.split(",")]:
    cfgs.append(("packed/lpw%d" % l, 0, l))
times = {c[0]: [] for c in cfgs}
b_line = hop * bps + nfft * 4
# warm the clocks: the chip needs ~10 back-to-back launches to reach its steady state
for _ in range(12):
    svc.compute_waterfall(iq, 0, nfft, dt, n_lines, hop=hop, window=args.window, out=out)
for rnd in range(args.rounds + 1):
    for name, v, l in cfgs:
        svc.set_option("force_generic", 1 if v is None else 0)
        svc.set_option("lines_per_wg", l)
        evs = []
        for rep in range(4):   # back-to-back, keep the last 3
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            svc.compute_waterfall(iq, 0, nfft, dt, n_lines, hop=hop, window=args.window, out=out)
            b.record(st)
            evs.append((a, b))
        torch.cuda.synchronize()
        if rnd == 0:
            err = (out - ref).abs().max().item()
            print("%-14s max|dB - generic| = %.3g" % (name, err), flush=True)
        else:
            times[name] += [a.elapsed_time(b) for a, b in evs[1:]]
for name, ts in times.items():
    med, mn = np.median(ts), np.min(ts)
    print("%-14s median %.3f ms  min %.3f ms  -> %.1f Mlines/s  %.0f GB/s (%.1f%% of 8 TB/s)" % (
        name, med, mn, n_lines / med / 1e3, n_lines * b_line / med / 1e6, n_lines * b_line / med / 1e6 / 80))
