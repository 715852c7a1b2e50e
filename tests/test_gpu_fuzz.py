"""Seeded randomised differential test of the whole dispatch (packed family, generic kernels,
four-step, EOF fill, staging) against the oracle: random datatype / nfft / hop / start / lines /
window / output format.  Sizes stay small so the CPU oracle finishes in seconds."""
import numpy as np
import pytest

import spectral_analyzer_amd as sa
from test_gpu_parity import check_fp32, check_fp64

pytestmark = pytest.mark.gpu
# one-off extended runs: SPEC_FUZZ_EXTRA_SEEDS=60 [SPEC_FUZZ_SEED_BASE=100] python -m pytest tests/test_gpu_fuzz.py -m gpu   (60 further seeds per test)
_BASE = int(__import__("os").environ.get("SPEC_FUZZ_SEED_BASE", "100"))
EXTRA_SEEDS = list(range(_BASE, _BASE + int(__import__("os").environ.get("SPEC_FUZZ_EXTRA_SEEDS", "0"))))
DTYPES = ["cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8", "cf64_le", "cf64_be"]


def _cases(seed, n):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        log2n = int(rng.choice([1, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]))
        nfft = 1 << log2n
        hop = int(rng.choice([nfft, nfft // 2 or 1, nfft // 4 or 1, int(rng.integers(1, 3 * nfft + 1)), 9 * nfft]))
        dt = str(rng.choice(DTYPES))
        budget = 1 << 17                                  # samples the oracle has to transform
        n_lines = int(rng.integers(1, max(2, min(300, budget // nfft))))
        if hop > 4 * nfft:
            n_lines = min(n_lines, 8)
        extra = int(rng.integers(0, 3))                   # lines past the end -> EOF fill
        start = int(rng.integers(0, 50))
        window = int(rng.integers(0, 2))
        fmt = int(rng.choice([sa.OUT_DB20_F32, sa.OUT_DB20_F32, sa.OUT_POW_F32, sa.OUT_DB20_F64]))
        on_device = bool(rng.integers(0, 2))
        lpw = int(rng.choice([0, 0, 1, 3, 50]))
        yield dt, nfft, hop, n_lines, extra, start, window, fmt, on_device, lpw, int(rng.integers(1, 1 << 30))


@pytest.mark.parametrize("seed", [1, 2, 3] + EXTRA_SEEDS)
def test_random_requests_match_oracle(svc, oracle, seed):
    import torch
    for dt, nfft, hop, n_lines, extra, start, window, fmt, on_device, lpw, s in _cases(seed, 60):
        bps = oracle.bytes_per_sample(dt)
        start_byte = start * bps
        n_samples = start + (n_lines - 1) * hop + nfft
        iq = oracle.synth_iq(dt, s, 0, n_samples)
        total = n_lines + extra
        tag = (dt, nfft, hop, n_lines, extra, start, window, fmt, on_device, lpw)
        svc.set_option("lines_per_wg", lpw)
        try:
            buf = torch.from_numpy(iq).cuda() if on_device else iq
            got = svc.compute_waterfall(buf, start_byte, nfft, dt, total, hop=hop, window=window, out_fmt=fmt)
            if on_device:
                torch.cuda.synchronize()
                got = got.cpu().numpy()
        finally:
            svc.set_option("lines_per_wg", 0)
        assert got.shape == (total, nfft), tag
        assert np.all(got[n_lines:] == -150.0), tag       # MC:994-998
        if fmt == sa.OUT_POW_F32:
            ref = oracle.waterfall(iq, start_byte, dt, nfft, hop, n_lines, window, power=True)
            tol = 4e-6 * max(np.log2(nfft), 2)
            assert np.abs(got[:n_lines] - ref).max() <= tol * ref.max(), tag
        else:
            ref = oracle.waterfall(iq, start_byte, dt, nfft, hop, n_lines, window)
            try:
                if fmt == sa.OUT_DB20_F64:
                    check_fp64(got[:n_lines], ref)
                else:
                    check_fp32(got[:n_lines], ref, max(nfft, 4))
            except AssertionError as e:
                raise AssertionError("%s: %s" % (tag, e))


def _burst_cases(seed, n):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        dt = str(rng.choice(DTYPES))
        count = int(rng.choice([1, 2, 63, 64, 65, 1000, 4097, int(rng.integers(1, 60000))]))
        start = int(rng.integers(0, 100))
        down = int(rng.choice([1, 2, 3, 7, 8, 16, 33, 100, 257]))
        fast = bool(rng.integers(0, 2))
        f_off = float(rng.choice([0.0, 0.25, -0.4999, 1e-7, float(rng.uniform(-0.5, 0.5))]))
        alpha = float(rng.choice([0.0, 1.0, 0.001, float(rng.uniform(0, 1))]))
        on_device = bool(rng.integers(0, 2))
        yield dt, start, count, down, fast, f_off, alpha, on_device, int(rng.integers(1, 1 << 30))


@pytest.mark.parametrize("seed", [11, 12] + EXTRA_SEEDS)
def test_random_burst_requests_match_oracle(svc, oracle, seed):
    """The burst chain (EDC:54-117, ADC:219-284) end to end on random requests: reader bit-exact, down-converter
    and traces within the fp64 tolerances of tests/test_gpu_burst.py, host and device residency."""
    import torch
    edc = sa.ExtractDownConvertService(svc)
    for dt, start, count, down, fast, f_off, alpha, on_device, s in _burst_cases(seed, 40):
        tag = (dt, start, count, down, fast, f_off, alpha, on_device)
        iq = oracle.synth_iq(dt, s, 0, start + count + 5)
        buf = torch.from_numpy(iq).cuda() if on_device else iq
        to_np = (lambda x: x.cpu().numpy()) if on_device else (lambda x: x)
        re, im = oracle.extract_iq(iq, start, count, dt)
        got = to_np(edc.extract_iq(buf, start, count, dt))
        assert np.array_equal(got[0], re) and np.array_equal(got[1], im), tag
        rr, ri = oracle.down_convert(re, im, f_off, down, 0 if fast else 1)
        d = edc.extract_and_down_convert(buf, start, count, dt, f_off, down, fast)
        dn = to_np(d)
        assert dn.shape == (2, count // down), tag
        if count // down == 0:
            continue
        m = max(np.abs(re).max(), np.abs(im).max(), 1e-30)
        assert np.abs(dn[0] - rr).max() <= 1e-12 * m and np.abs(dn[1] - ri).max() <= 1e-12 * m, tag
        fs = 1e6 / down
        mag, frq = to_np(svc.magnitude_trace(d, alpha)), to_np(svc.inst_freq_trace(d, alpha, fs, 1e9))
        ref_mag = oracle.magnitude_trace(dn[0], dn[1], alpha)
        v = 10 ** (ref_mag / 20)
        ok = np.isfinite(ref_mag) & (v >= 1e-9 * v[np.isfinite(v)].max()) if np.isfinite(v).any() else np.zeros(len(v), bool)
        assert np.abs(mag[ok] - ref_mag[ok]).max(initial=0.0) <= 1e-9, tag
        ref_frq = oracle.inst_freq_trace(dn[0], dn[1], alpha, fs, 1e9)
        # A phase step of exactly pi between two samples (quantised samples, no mixer: x[n] = -c x[n-1]) is wrapped to +fs/2 or
        # -fs/2 by the last bit of two atan2 results (ADC:268-272 compares their difference with pi): the reference's own value is
        # then a property of its libm, not of its source, and the smoothed trace carries the choice on.  Such requests (found by the
        # extended random runs: 1 in ~20 000) are not compared.
        step = np.abs(np.diff(np.arctan2(dn[1], dn[0])))
        if np.any(np.abs(step - np.pi) < 1e-9):
            continue
        assert frq.shape == ref_frq.shape and np.abs(frq - ref_frq).max(initial=0.0) <= 1e-9 * fs + 1e-6, tag


def _welch_cases(seed, n):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        nfft = 1 << int(rng.choice([4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]))
        hop = int(rng.choice([nfft, nfft // 2, nfft // 4, int(rng.integers(1, 2 * nfft + 1))]))
        n_seg = int(rng.integers(1, max(2, min(40, (1 << 17) // nfft))))
        n_psd = int(rng.integers(1, 4))
        yield (str(rng.choice(DTYPES)), nfft, hop, n_seg, n_psd, int(rng.integers(0, 30)), int(rng.integers(0, 2)),
               int(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.integers(1, 1 << 30)))


@pytest.mark.parametrize("seed", [4, 5, 6] + EXTRA_SEEDS)
def test_random_welch_requests_match_oracle(svc, oracle, seed):
    """The Welch entry (ADC:303-313 call shape, batched) over random datatype / nfft / hop / segments / batch / start /
    window / scaling, host or device input: linear within 5e-6 of the PSD's peak (fp32 sums; 1e-9 for the fp64 family that
    cf64 and big-endian inputs take), dB within 2e-3 on bins above 1e-3 of the peak."""
    import torch
    fs = 1.0e6
    for dt, nfft, hop, n_seg, n_psd, start, window, scaling, on_device, s in _welch_cases(seed, 40):
        bps = oracle.bytes_per_sample(dt)
        per = (n_seg - 1) * hop + nfft + int(s % 7)          # a few samples between the PSDs' spans
        iq = oracle.synth_iq(dt, s, 0, start + per * n_psd)
        tag = (dt, nfft, hop, n_seg, n_psd, start, window, scaling, on_device)
        buf = torch.from_numpy(iq).cuda() if on_device else iq
        res = []
        for db in (False, True):
            _, p = svc.welch_psd(buf, start * bps, dt, fs, nfft=nfft, hop=hop, n_seg=n_seg, window=window, scaling=scaling,
                                 n_psd=n_psd, psd_stride_bytes=per * bps, db=db)
            if on_device:
                torch.cuda.synchronize()
                p = p.cpu().numpy()
            res.append(np.asarray(p, dtype=np.float64))
        assert res[0].shape == (n_psd, nfft), tag
        for b in range(n_psd):
            _, ref = oracle.welch_psd(iq, (start + b * per) * bps, dt, nfft, hop, n_seg, window, scaling, fs)
            _, ref_db = oracle.welch_psd(iq, (start + b * per) * bps, dt, nfft, hop, n_seg, window, scaling, fs, db=True)
            assert np.abs(res[0][b] - ref).max() <= 5e-6 * ref.max(), tag
            strong = ref >= 1e-3 * ref.max()
            assert np.abs(res[1][b] - ref_db)[strong].max() <= 2e-3, tag


@pytest.mark.parametrize("seed", [10, 11] + EXTRA_SEEDS)
def test_random_redraws_match_the_restated_renderer(svc, oracle, seed):
    """One redraw (MC:980-999 lines, MC:1261-1291 image) on random requests: the fused form (compact tile) gives the very
    pixels of the two-pass form, and both are the restated Java renderer applied to the GPU's own dB tile -- any datatype,
    size, hop, canvas, dB range and colour map, columns past the end of the buffer included."""
    rng = np.random.default_rng(seed)
    for _ in range(30):
        dt = str(rng.choice(DTYPES))
        nfft = 1 << int(rng.choice([6, 7, 8, 9, 10, 11, 12, 13]))
        hop = int(rng.choice([nfft, nfft // 2, int(rng.integers(1, 2 * nfft + 1))]))
        width = int(rng.integers(1, 120))
        height = int(rng.choice([1, int(rng.integers(2, nfft + 1)), int(rng.integers(nfft, 2 * nfft + 2))]))
        past = int(rng.integers(0, 4))                      # columns that run past the end: -150 dB
        n = max(width - past - 1, 0) * hop + nfft
        iq = oracle.synth_iq(dt, int(rng.integers(1, 1 << 30)), 0, n)
        fs, lo = float(rng.uniform(1e3, 1e8)), float(rng.uniform(-160.0, -40.0))
        hi = lo + float(rng.uniform(1.0, 120.0))
        cmap = int(rng.choice([sa.CMAP_GRAYSCALE, sa.CMAP_HEATMAP]))
        tag = (dt, nfft, hop, width, height, past, fs, lo, hi, cmap)
        fused = svc.waterfall_render(iq, 0, nfft, dt, width, height, fs, min_db=lo, max_db=hi, colormap=cmap, hop=hop)
        svc.set_option("render_fused", 0)
        try:
            plain = svc.waterfall_render(iq, 0, nfft, dt, width, height, fs, min_db=lo, max_db=hi, colormap=cmap, hop=hop)
        finally:
            svc.set_option("render_fused", 1)
        assert fused.shape == (height, width, 4) and np.array_equal(fused, plain), tag
        tile = svc.compute_waterfall(iq, 0, nfft, dt, width, hop=hop)
        assert np.array_equal(fused, oracle.render_spectrogram(tile.astype(np.float64), height, fs, lo, hi, cmap)), tag


@pytest.mark.parametrize("seed", [12, 13] + EXTRA_SEEDS)
def test_random_sharded_requests_equal_the_single_context_tile(svc, oracle, seed):
    """spec_waterfall_multi over one to four contexts (all on device 0 on the one-GPU box) on random requests -- datatype, size,
    hop, start, line count (also fewer lines than contexts), lines past the end, window, output format, host or device tile,
    number of pieces -- is the single-context tile bit for bit."""
    import torch
    rng = np.random.default_rng(seed)
    peers = [sa.SpectralService(0, stream=torch.cuda.Stream().cuda_stream) for _ in range(3)]
    try:
        for _ in range(25):
            dt = str(rng.choice(DTYPES))
            nfft = 1 << int(rng.choice([6, 8, 9, 10, 11, 12, 13]))
            hop = int(rng.choice([nfft, nfft // 2, nfft // 4, int(rng.integers(1, 2 * nfft + 1))]))
            n_lines = int(rng.choice([1, 2, 3, int(rng.integers(4, max(5, min(400, (1 << 18) // nfft))))]))
            extra, start = int(rng.integers(0, 3)), int(rng.integers(0, 20))
            window = int(rng.integers(0, 2))
            fmt = int(rng.choice([sa.OUT_DB20_F32, sa.OUT_POW_F32, sa.OUT_DB20_F64]))
            n_ctx = int(rng.integers(1, 5))
            bps = oracle.bytes_per_sample(dt)
            iq = oracle.synth_iq(dt, int(rng.integers(1, 1 << 30)), 0, start + (n_lines - 1) * hop + nfft)
            tag = (dt, nfft, hop, n_lines, extra, start, window, fmt, n_ctx)
            one = svc.compute_waterfall(iq, start * bps, nfft, dt, n_lines + extra, hop=hop, window=window, out_fmt=fmt)
            services = [svc] + peers[:n_ctx - 1]
            if rng.integers(0, 2):
                got = sa.compute_waterfall_multi(services, iq, start * bps, nfft, dt, n_lines + extra, hop=hop, window=window, out_fmt=fmt)
            else:
                out = torch.full((n_lines + extra, nfft), float("nan"), dtype=torch.from_numpy(one[:1]).dtype, device="cuda")
                sa.compute_waterfall_multi(services, iq, start * bps, nfft, dt, n_lines + extra, hop=hop, window=window, out_fmt=fmt,
                                           out=out, n_chunks=int(rng.integers(0, 7)))
                got = out.cpu().numpy()
            assert got.dtype == one.dtype and np.array_equal(got, one), tag
            # a batch of PSDs over the same contexts (spec_welch_psd_multi): small batches take the same form of the kernel in
            # every context, so the single-context batch comes back bit for bit
            n_psd, n_seg = int(rng.integers(1, 9)), int(rng.integers(1, 6))
            wn = 1 << int(rng.choice([4, 8, 10, 12]))
            whop = int(rng.choice([wn, wn // 2, int(rng.integers(1, wn + 1))]))
            per = (n_seg - 1) * whop + wn + int(rng.integers(0, 9))
            wiq = oracle.synth_iq(dt, int(rng.integers(1, 1 << 30)), 0, start + per * n_psd)
            db = bool(rng.integers(0, 2))
            f1, p1 = svc.welch_psd(wiq, start * bps, dt, 1e6, nfft=wn, hop=whop, n_seg=n_seg, window=window, n_psd=n_psd,
                                   psd_stride_bytes=per * bps, db=db)
            f2, p2 = sa.welch_psd_multi(services, wiq, start * bps, dt, 1e6, wn, whop, n_seg, n_psd, per * bps, window=window, db=db)
            assert np.array_equal(f1, f2) and np.array_equal(p1, p2), tag + (n_psd, n_seg, wn, whop, db)
    finally:
        for p in peers:
            p.close()


@pytest.mark.parametrize("seed", [16, 17] + EXTRA_SEEDS)
def test_random_psd_dialog_calls_match_scipy(svc, seed):
    """calculatePsdWelch(double[2][N], fs, nfft) (ADC:303-313) on random bursts: nfft = the burst length for short bursts (the
    dialog's rule, any integer) or any length up to 8192, explicit hop; against scipy.signal.welch -- an oracle that shares no
    code with the build -- to 1e-9 of the PSD's peak (fp64 end to end), host and device input."""
    import scipy.signal
    import torch
    rng = np.random.default_rng(seed)
    for _ in range(14):
        n = int(rng.choice([2, 3, int(rng.integers(4, 600)), int(rng.integers(600, 8192)), int(rng.integers(8192, 40000))]))
        nfft = n if (n < 8192 and rng.integers(0, 2)) else int(rng.choice([min(n, 8192), int(rng.integers(2, min(n, 8192) + 1)),
                                                                         1 << int(rng.integers(1, int(np.log2(min(n, 8192))) + 1))]))
        hop = int(rng.choice([max(nfft // 2, 1), int(rng.integers(1, nfft + 1))]))
        fs = float(rng.uniform(1e3, 1e8))
        scaling = int(rng.integers(0, 2))
        x = rng.normal(size=n) + 1j * rng.normal(size=n) + 2 * np.exp(2j * np.pi * float(rng.uniform(-0.5, 0.5)) * np.arange(n))
        data = np.stack([x.real, x.imag])
        tag = (n, nfft, hop, fs, scaling)
        arg = torch.from_numpy(data).cuda() if rng.integers(0, 2) else data
        out = svc.calculate_psd_welch(arg, fs, nfft, hop=hop, scaling=scaling, db=False)
        f_ref, p_ref = scipy.signal.welch(x, fs, window="hann", nperseg=nfft, noverlap=nfft - hop, nfft=nfft, detrend=False,
                                          return_onesided=False, scaling="density" if scaling == sa.PSD_DENSITY else "spectrum")
        assert out.shape == (2, nfft), tag
        assert np.allclose(out[0], np.fft.fftshift(f_ref), rtol=1e-12, atol=1e-9 * fs), tag
        assert np.abs(out[1] - np.fft.fftshift(p_ref)).max() <= 1e-9 * p_ref.max(), tag
