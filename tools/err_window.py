#!/usr/bin/env python3
"""fp32 error tiers of windowed long lines, computed window (spec_k_v2h.hip) against table window (four-step path, family): the worst
dB error on bins >= 1e-3 M and >= 1e-4 M and the worst linear error over many random lines.  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
from oracle import spec_oracle as so
so.build()
svc = sa.SpectralService(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for nfft, knob in ((32768, "large_single"), (16384, "mid_single"), (8192, "small_single")):
    for dt in ("cf32_le", "cf32_be", "ci16_le"):
        worst = {0: [0, 0, 0], 1: [0, 0, 0]}
        for it in range(24):
            n_lines = 8
            iq = so.synth_iq(dt, int(rng.integers(1, 1 << 30)), 0, (n_lines - 1) * nfft + nfft)
            ref = so.waterfall(iq, 0, dt, nfft, nfft, n_lines, 1)
            mag_r = 10.0 ** (ref / 20.0); M = mag_r.max(axis=1, keepdims=True)
            d = torch.from_numpy(iq).cuda()
            for mode in (0, 1):
                svc.set_option(knob, mode)
                got = svc.compute_waterfall(d, 0, nfft, dt, n_lines, hop=nfft, window=1).cpu().numpy().astype(np.float64)
                err = np.abs(got - ref)
                lin = (np.abs(10.0 ** (got / 20.0) - mag_r) / (M * np.log2(nfft))).max()
                w = worst[mode]
                w[0] = max(w[0], err[mag_r >= 1e-3 * M].max()); w[1] = max(w[1], err[mag_r >= 1e-4 * M].max()); w[2] = max(w[2], lin)
        svc.set_option(knob, 2 if knob != "large_single" else 1)
        for mode, name in ((0, "table window (family / four-step)"), (1, "computed window (half-line kernel)")):
            print("%5d %-8s %-36s dB @1e-3 M %.2e  @1e-4 M %.2e  linear %.2e M log2N" % (nfft, dt, name, *worst[mode]), flush=True)
