#!/usr/bin/env python3
"""Extended randomised run of the single-workgroup and pair kernels (spec_k_v2h.hip at 8192 / 16384 / 32768 points, spec_k_v2q.hip at
65536, spec_k_v3h.hip at 16384 and -- v3q_kernel -- 32768 points in fp64) against the oracle: formats, hops, start bytes, windows, output formats, lines per workgroup, the
dispatch knobs, host and device buffers, repeated calls (a result that changes between two calls is a race).  One-off
development run (tests/test_gpu_fuzz.py is the suite's share of it).   python tools/fuzz_large.py [requests=400] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import spectral_analyzer_amd as sa
from oracle import spec_oracle as so
from test_gpu_parity import check_fp32, check_fp64, fp64_pow_tol
so.build()
svc = sa.SpectralService(0)
DT32 = ["cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8"]
DT64 = DT32 + ["cf64_le", "cf64_be"]
n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
n_cases = {}
for it in range(n_req):
    f64 = bool(rng.integers(0, 3) == 0)
    log2n = int(rng.choice([14, 15])) if f64 else int(rng.choice([13, 14, 15, 16]))   # (round 5: the pair kernels -- 65536 points fp32, 32768 points fp64)
    nfft = 1 << log2n
    hop = int(rng.choice([nfft, nfft // 2, nfft // 4, int(rng.integers(1, 3 * nfft + 1)), 7 * nfft]))
    dt = str(rng.choice(DT64 if f64 else DT32))
    n_lines = int(rng.integers(1, 9)) if hop <= 2 * nfft else int(rng.integers(1, 4))
    extra = int(rng.integers(0, 3)); start = int(rng.integers(0, 50)); window = int(rng.integers(0, 2))
    if f64:
        fmt = int(rng.choice([sa.OUT_DB20_F64, sa.OUT_POW_F64] + ([sa.OUT_DB20_F32] if dt.startswith("cf64") else [])))
    else:
        fmt = int(rng.choice([sa.OUT_DB20_F32, sa.OUT_POW_F32]))
    lpw = int(rng.choice([0, 0, 1, 3, 50])); mid = int(rng.choice([2, 2, 1, 0])); small = int(rng.choice([2, 1, 1, 0]))
    pair = int(rng.choice([1, 2, 2])); ilv = int(rng.integers(0, 2))
    bps = so.bytes_per_sample(dt)
    iq = so.synth_iq(dt, int(rng.integers(1, 1 << 30)), 0, start + (n_lines - 1) * hop + nfft)
    svc.set_option("lines_per_wg", lpw); svc.set_option("mid_single", mid); svc.set_option("small_single", small)
    svc.set_option("large_pair", pair); svc.set_option("pair_interleave", ilv)
    dev = bool(rng.integers(0, 2))
    buf = torch.from_numpy(iq).cuda() if dev else iq
    outs = []
    for rep in range(2):
        got = svc.compute_waterfall(buf, start * bps, nfft, dt, n_lines + extra, hop=hop, window=window, out_fmt=fmt)
        if dev:
            torch.cuda.synchronize(); got = got.cpu().numpy()
        outs.append(got)
    tag = (dt, nfft, hop, n_lines, extra, start, window, fmt, lpw, mid, small, pair, ilv, dev)
    assert np.array_equal(outs[0], outs[1], equal_nan=True), ("two calls differ", tag)
    got = outs[0]
    assert np.all(got[n_lines:] == -150.0), tag
    try:
        if fmt in (sa.OUT_POW_F32, sa.OUT_POW_F64):
            ref = so.waterfall(iq, start * bps, dt, nfft, hop, n_lines, window, power=True)
            tol = fp64_pow_tol(nfft) if fmt == sa.OUT_POW_F64 else 4e-6 * np.log2(nfft)
            assert np.abs(got[:n_lines] - ref).max() <= tol * ref.max()
        else:
            ref = so.waterfall(iq, start * bps, dt, nfft, hop, n_lines, window)
            if fmt == sa.OUT_DB20_F64: check_fp64(got[:n_lines], ref)
            elif f64: assert np.abs(got[:n_lines] - ref).max() <= 2e-5
            else: check_fp32(got[:n_lines], ref, nfft)
    except AssertionError as e:
        raise AssertionError("%s: %s" % (tag, e))
    key = (nfft, "fp64" if f64 else "fp32")
    n_cases[key] = n_cases.get(key, 0) + 1
print("extended randomised run green:", ", ".join("%d requests of %d points (%s)" % (v, k[0], k[1]) for k, v in sorted(n_cases.items())))
