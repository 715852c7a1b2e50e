import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, spectral_analyzer_amd as sa
from oracle import spec_oracle as so
svc = sa.SpectralService(0)
rng = np.random.default_rng(1)
for n in (20000, 200000, 2000000, 20000000):
    data = rng.standard_normal((2, n))
    for _ in range(3): svc.calculate_psd_welch(data, 1e6, 8192)
    t0 = time.perf_counter()
    for _ in range(5): f, p = svc.calculate_psd_welch(data, 1e6, 8192)
    g = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    raw = np.empty(2 * n); raw[0::2], raw[1::2] = data[0], data[1]
    so.welch_psd(raw.view(np.uint8), 0, "cf64_le", 8192, 4096, (n - 8192) // 4096 + 1, fs=1e6)
    c = time.perf_counter() - t0
    print("calculatePsdWelch(double[2][%d], fs, 8192): GPU %.2f ms   C oracle (1 thread) %.2f ms" % (n, g * 1e3, c * 1e3), flush=True)
