import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
from oracle import spec_oracle as so
svc = sa.SpectralService(0)
dt, nfft, hop, n_lines = "cf32_le", 4096, 2048, 9
iq = so.synth_iq(dt, 5, 0, (n_lines - 1) * hop + nfft)
ref = so.waterfall(iq, 0, dt, nfft, hop, n_lines)
for lpw in (0, 1, 2, 3):
    svc.set_option("lines_per_wg", lpw)
    got = svc.compute_waterfall(iq, 0, nfft, dt, n_lines, hop=hop).astype(np.float64)
    e = np.abs(10 ** (got / 20) - 10 ** (ref / 20)) / (10 ** (ref / 20)).max()
    print("lpw", lpw, "per-line max err:", np.array2string(e.max(axis=1), precision=2))
    bad = np.argwhere(e > 1e-4)
    print("  n bad", len(bad), "first bad", bad[:5].tolist(), "bins of bad in line:", sorted(set((bad[:, 1] // 256).tolist()))[:16])
