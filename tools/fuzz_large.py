import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import spectral_analyzer_amd as sa
from oracle import spec_oracle as so
from test_gpu_parity import check_fp32
so.build()
svc = sa.SpectralService(0)
DT = ["cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8"]
rng = np.random.default_rng(20261004)
n_cases = 0
for it in range(260):
    log2n = int(rng.choice([14, 15]))
    nfft = 1 << log2n
    hop = int(rng.choice([nfft, nfft // 2, nfft // 4, int(rng.integers(1, 3 * nfft + 1)), 7 * nfft]))
    dt = str(rng.choice(DT))
    n_lines = int(rng.integers(1, 9)) if hop <= 2 * nfft else int(rng.integers(1, 4))
    extra = int(rng.integers(0, 3)); start = int(rng.integers(0, 50)); window = int(rng.integers(0, 2))
    fmt = int(rng.choice([sa.OUT_DB20_F32, sa.OUT_POW_F32]))
    lpw = int(rng.choice([0, 0, 1, 3, 50])); mid = int(rng.choice([2, 2, 1, 0]))
    bps = so.bytes_per_sample(dt)
    iq = so.synth_iq(dt, int(rng.integers(1, 1 << 30)), 0, start + (n_lines - 1) * hop + nfft)
    svc.set_option("lines_per_wg", lpw); svc.set_option("mid_single", mid)
    dev = bool(rng.integers(0, 2))
    got = svc.compute_waterfall(torch.from_numpy(iq).cuda() if dev else iq, start * bps, nfft, dt, n_lines + extra, hop=hop, window=window, out_fmt=fmt)
    if dev:
        torch.cuda.synchronize(); got = got.cpu().numpy()
    tag = (dt, nfft, hop, n_lines, extra, start, window, fmt, lpw, mid, dev)
    assert np.all(got[n_lines:] == -150.0), tag
    if fmt == sa.OUT_POW_F32:
        ref = so.waterfall(iq, start * bps, dt, nfft, hop, n_lines, window, power=True)
        assert np.abs(got[:n_lines] - ref).max() <= 4e-6 * np.log2(nfft) * ref.max(), tag
    else:
        ref = so.waterfall(iq, start * bps, dt, nfft, hop, n_lines, window)
        try: check_fp32(got[:n_lines], ref, nfft)
        except AssertionError as e: raise AssertionError("%s: %s" % (tag, e))
    n_cases += 1
print("extended randomised run: %d requests of 16384 / 32768 points green" % n_cases)
