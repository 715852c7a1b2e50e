#!/usr/bin/env python3
"""computeMagnitudes (SS:33-85: one fp64 line per call) at 16384 points: the single-workgroup kernel against the two-launch
four-step path it replaces for short calls ("large_single" = 0).  Development tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
svc = sa.SpectralService(0)
N = 16384
for dt in ("cf32_le", "ci16_le", "cf64_le"):
    iq = np.frombuffer(svc.synth_iq(dt, 3, 0, 4 * N).cpu().numpy().tobytes(), np.uint8)
    for single in (0, 1):
        svc.set_option("large_single", single)
        for _ in range(50): svc.compute_magnitudes(iq, 0, N, dt)
        t0 = time.perf_counter()
        for _ in range(300): svc.compute_magnitudes(iq, 0, N, dt)
        us = (time.perf_counter() - t0) / 300 * 1e6
        d = torch.from_numpy(iq).cuda()
        for n_lines in (1, 8, 63):
            for _ in range(20): svc.compute_waterfall(d, 0, N, dt, n_lines, hop=N // 4 if n_lines > 4 else N, out_fmt=sa.OUT_DB20_F64)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): svc.compute_waterfall(d, 0, N, dt, n_lines, hop=N // 4 if n_lines > 4 else N, out_fmt=sa.OUT_DB20_F64)
            torch.cuda.synchronize()
            print("%-8s large_single %d: %2d resident lines per call %.1f us" % (dt, single, n_lines, (time.perf_counter() - t0) / 200 * 1e6))
        print("%-8s large_single %d: computeMagnitudes (host buffer, double[] back) %.1f us/call" % (dt, single, us), flush=True)
svc.set_option("large_single", 1)
