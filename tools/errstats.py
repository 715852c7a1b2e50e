import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectral_analyzer_amd as sa
from oracle import spec_oracle as so
svc = sa.SpectralService(0)
for dt in ["cf32_le", "ci16_le", "cu8"]:
  for nfft in (1024, 4096, 16384):
    hop, n_lines = nfft // 4, 41
    iq = so.synth_iq(dt, 9, 0, (n_lines - 1) * hop + nfft)
    ref = so.waterfall(iq, 0, dt, nfft, hop, n_lines)
    for gen in (0, 1):
        svc.set_option("force_generic", gen)
        got = svc.compute_waterfall(iq, 0, nfft, dt, n_lines, hop=hop).astype(np.float64)
        mg, mr = 10 ** (got / 20), 10 ** (ref / 20)
        M = mr.max(axis=1, keepdims=True)
        x = so.np_decode(iq, 0, (n_lines - 1) * hop + nfft, dt)
        l2 = np.sqrt(np.mean(np.abs(x) ** 2) * nfft)
        e = np.abs(mg - mr)
        eps = 2.0 ** -24
        print(dt, nfft, "generic" if gen else "auto", "max abs err %.3g  rms %.3g | /(eps*sqrt(log2N)*||x||2): max %.2f rms %.2f | /(M log2N) max %.3g | dB err max: >=1e-4M %.3g  >=1e-3M %.3g >=1e-2M %.3g" % (
            e.max(), np.sqrt((e**2).mean()), e.max() / (eps * np.sqrt(np.log2(nfft)) * l2), np.sqrt((e**2).mean()) / (eps * np.sqrt(np.log2(nfft)) * l2),
            (e / (M * np.log2(nfft))).max(), np.abs(got - ref)[mr >= 1e-4 * M].max(), np.abs(got - ref)[mr >= 1e-3 * M].max(), np.abs(got - ref)[mr >= 1e-2 * M].max()))
