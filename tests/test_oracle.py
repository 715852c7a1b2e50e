"""CPU tests of the oracle (no GPU): the analytic known-answer tests of SURVEY.md
8(c) derived from SpectralService.java:40-82, the cross-check of the C
restatement against numpy.fft / scipy.signal.welch, and the committed golden
fixtures.  The reference publishes no vectors of its own (its only test is a
Spring context load), so these KATs are what pins the oracle."""
import os

import numpy as np
import pytest
import scipy.signal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DTYPES = ["cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8", "cf64_le", "cf64_be"]


def cf32_bytes(x, be=False):
    v = np.empty(2 * len(x), dtype=">f4" if be else "<f4")
    v[0::2], v[1::2] = x.real, x.imag
    return v.view(np.uint8)


def test_k1_all_zero_is_minus_200(oracle):
    # SS:81: 20*log10(0 + 1e-10) = -200 exactly
    for n in (64, 1024):
        out = oracle.compute_magnitudes(np.zeros(8 * n, np.uint8), 0, n, "cf32_le")
        assert np.all(out == -200.0)


def test_k2_unknown_datatype_and_cf64_defect(oracle):
    # SS:60-63: unknown datatype decodes to zeros; cf64 has no branch in the service (SS:35-38)
    buf = oracle.synth_iq("cf64_le", 1, 0, 256)
    assert np.all(oracle.compute_magnitudes(buf, 0, 256, "xyz") == -200.0)
    assert np.all(oracle.compute_magnitudes(buf, 0, 256, "cf64_le", cf64_decode=False) == -200.0)
    assert oracle.compute_magnitudes(buf, 0, 256, "cf64_le", cf64_decode=True).max() > 0


def test_k3_unit_impulse_is_flat(oracle):
    n = 256
    x = np.zeros(n, complex); x[0] = 1
    out = oracle.compute_magnitudes(cf32_bytes(x), 0, n, "cf32_le")
    assert np.allclose(out, 20 * np.log10(1 + 1e-10), atol=1e-12)


def test_k4_dc_lands_at_n_over_2(oracle):
    n = 512
    out = oracle.compute_magnitudes(cf32_bytes(np.ones(n, complex)), 0, n, "cf32_le")
    assert out[n // 2] == pytest.approx(20 * np.log10(n + 1e-10), abs=1e-12)   # SS:78
    lin = 10 ** (np.delete(out, n // 2) / 20)
    assert lin.max() < 1e-9            # exact zeros up to the +1e-10 epsilon / round-off


@pytest.mark.parametrize("k", [1, 37, 255, 256, 300, 511])
def test_k5_complex_tone_bin(oracle, k):
    n = 512
    x = np.exp(2j * np.pi * k * np.arange(n) / n)
    out = oracle.compute_magnitudes(cf32_bytes(x), 0, n, "cf32_le")
    assert int(np.argmax(out)) == (k + n // 2) % n
    assert out.max() == pytest.approx(20 * np.log10(n), abs=1e-4)   # cf32 rounding of the tone


def test_k6_decode_table(oracle):
    # SS:44-45 ci16 / 32768 ; SS:51-54 (b & 0xFF - 127.5) / 128 ; SS:56-59 b / 128
    ci16 = np.array([0x8000, 0x7FFF], dtype="<u2").view(np.uint8)
    z = oracle.np_decode(ci16, 0, 1, "ci16_le")[0]
    assert (z.real, z.imag) == (-1.0, 32767 / 32768)
    z = oracle.np_decode(np.array([0, 255], np.uint8), 0, 1, "cu8")[0]
    assert (z.real, z.imag) == (-0.99609375, 0.99609375)
    z = oracle.np_decode(np.array([0x80, 0x7F], np.uint8), 0, 1, "ci8")[0]
    assert (z.real, z.imag) == (-1.0, 0.9921875)
    # and the C decode agrees through a 1-point... use a 2-point FFT: X0 = a + b, X1 = a - b
    for dt, raw in (("ci16_le", np.array([0x8000, 0x7FFF, 0x0001, 0xFFFF], dtype="<u2").view(np.uint8)),
                    ("cu8", np.array([0, 255, 128, 127], np.uint8)), ("ci8", np.array([0x80, 0x7F, 1, 0xFF], np.uint8))):
        x = oracle.np_decode(raw, 0, 2, dt)
        ref = 20 * np.log10(np.abs(np.array([x[0] - x[1], x[0] + x[1]])) + 1e-10)   # fftshifted
        assert np.allclose(oracle.compute_magnitudes(raw, 0, 2, dt), ref, atol=1e-12)


@pytest.mark.parametrize("base", ["cf32", "ci16", "cf64"])
def test_k7_big_and_little_endian_files_agree(oracle, base):
    le = oracle.synth_iq(base + "_le", 5, 10, 2048)
    be = oracle.synth_iq(base + "_be", 5, 10, 2048)
    assert not np.array_equal(le, be)
    a = oracle.waterfall(le, 0, base + "_le", 512, 256, 7)
    b = oracle.waterfall(be, 0, base + "_be", 512, 256, 7)
    assert np.array_equal(a, b)


def test_k8_eof_lines_are_minus_150(oracle):
    # MC:987-998: a line whose last byte passes the capacity is filled with -150.0
    buf = oracle.synth_iq("ci16_le", 2, 0, 1000)
    out = oracle.waterfall(buf, 0, "ci16_le", 256, 256, 5)
    assert oracle.count_lines(buf.size, 0, "ci16_le", 256, 256) == 3
    assert np.all(out[3:] == -150.0) and np.all(out[:3] > -150.0)
    # start offset moves the boundary
    assert oracle.count_lines(buf.size, 4 * 300, "ci16_le", 256, 256) == 2
    assert oracle.count_lines(buf.size, buf.size, "ci16_le", 256, 256) == 0


@pytest.mark.parametrize("datatype", DTYPES)
def test_k9_parseval(oracle, datatype):
    n, lines = 1024, 3
    buf = oracle.synth_iq(datatype, 9, 0, n * lines)
    p = oracle.waterfall(buf, 0, datatype, n, n, lines, power=True)
    x = oracle.np_decode(buf, 0, n * lines, datatype).reshape(lines, n)
    assert np.allclose(p.sum(axis=1), n * (np.abs(x) ** 2).sum(axis=1), rtol=1e-12)


def cm3_lin_bound(nfft):
    """What the reference's own transform may be off by, relative to the line's peak magnitude M: its twiddles are
    running products (commons-math3 FastFourierTransformer: wSubN0ToR *= wSubN0), so twiddle r of a stage carries
    about r eps of error.  Measured here (test below prints it): <= 1.5e-17 N on Gaussian input, <= 0.9e-17 N on the
    synthetic recordings; the bound leaves a factor of about three."""
    return 4e-17 * nfft


@pytest.mark.parametrize("datatype", DTYPES)
@pytest.mark.parametrize("nfft", [2, 4, 8, 64, 1024, 4096, 16384, 65536])
def test_k10_both_transforms_match_numpy_fft(oracle, datatype, nfft):
    """The published commons-math3 algorithm (FFT_CM3: what the reference computes) and the exact-twiddle yardstick
    (FFT_EXACT) against numpy.fft on the same bytes.  The yardstick holds 1e-12 of the peak POWER at every length; the
    reference's transform holds its own N-dependent bound -- its error is the reference's, reproduced, not the oracle's."""
    hop = max(1, nfft // 2)
    lines = 3
    buf = oracle.synth_iq(datatype, nfft, 7, (lines - 1) * hop + nfft)
    for window in (oracle.WIN_RECT, oracle.WIN_HANN):
        X = oracle.np_spectrum(buf, 0, datatype, nfft, hop, lines, window)
        b = np.fft.fftshift(np.abs(X) ** 2, axes=1)
        e = oracle.waterfall(buf, 0, datatype, nfft, hop, lines, window, power=True, fft=oracle.FFT_EXACT)
        assert np.abs(e - b).max() <= 1e-12 * b.max()
        a = oracle.waterfall(buf, 0, datatype, nfft, hop, lines, window, power=True, fft=oracle.FFT_CM3)
        assert np.abs(a - b).max() <= max(1e-12, 2 * cm3_lin_bound(nfft)) * b.max()      # power: twice the linear error


def test_cm3_transform_own_error_per_length(oracle, capsys):
    """Prints what the reference's transform is off by, per length, against numpy.fft and against the exact-twiddle
    transform -- the figure the fp64 tolerance of the GPU parity tests is stated against (tests/test_gpu_parity.py
    fp64_tol) -- and holds it under cm3_lin_bound."""
    rng = np.random.default_rng(11)
    rows = []
    for log2n in range(1, 17):
        n = 1 << log2n
        x = rng.normal(size=n) + 1j * rng.normal(size=n) + 30 * np.exp(2j * np.pi * 0.123 * np.arange(n))
        ref = np.fft.fft(x)
        M = np.abs(ref).max()
        cm3, exact = oracle.fft_forward(x, oracle.FFT_CM3), oracle.fft_forward(x, oracle.FFT_EXACT)
        e_cm3, e_exact = np.abs(cm3 - ref).max() / M, np.abs(exact - ref).max() / M
        rows.append((n, e_cm3, e_exact))
        assert e_exact <= 2e-15, (n, e_exact)
        assert e_cm3 <= max(2e-15, cm3_lin_bound(n)), (n, e_cm3)
    with capsys.disabled():
        print("\n  N      cm3 (reference) err / M    exact-twiddle err / M")
        for n, a, b in rows:
            print("  %-6d %-26.3g %.3g" % (n, a, b))
    assert rows[-1][1] > 20 * rows[-1][2]        # at 65536 points the recurrence, not rounding, is what one sees


def test_cm3_root_tables_and_first_stages(oracle):
    """What is checkable about the restated library without the jar: (i) the stage roots are the published literals
    (tools/gen_cm3_roots.py asserts the ones known from the source and regenerates the rest); (ii) lengths 1, 2 and 4 use
    no multiplication at all, so they are exact on small integers; (iii) length 8 uses the root 0x1.6a09e667f3bcdp-1 -
    0x1.6a09e667f3bccp-1 i (cos and sin of the DOUBLE pi/4 differ in the last bit), visible in X[1] of an impulse at n = 1."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_cm3_roots.py")], stdout=subprocess.DEVNULL)
    for n in (1, 2, 4):
        x = (np.arange(n) + 1) + 1j * (np.arange(n) * 3 - 2)
        assert np.array_equal(oracle.fft_forward(x), np.fft.fft(x))
    x = np.zeros(8, complex); x[1] = 1
    X = oracle.fft_forward(x)
    assert X[1].real == float.fromhex("0x1.6a09e667f3bcdp-1") and X[1].imag == -float.fromhex("0x1.6a09e667f3bccp-1")
    # the fingerprint of the recurrence: W_8^2 is NOT the exact -i but the running product (c - s i)^2 =
    # (c c - s s) - (c s + s c) i with c != s in the last bit -> a real part of 2^-52 where an exact table has 0
    c, sn = float.fromhex("0x1.6a09e667f3bcdp-1"), float.fromhex("0x1.6a09e667f3bccp-1")
    assert X[2].real == c * c - sn * sn == 2.0 ** -52 and X[2].imag == c * -sn + -sn * c
    e2 = oracle.fft_forward(x, oracle.FFT_EXACT)[2]
    assert abs(e2.real) < 1e-19 and e2.imag == -1.0           # cos(pi/2) in long double, rounded once
    assert X[0] == 1 and X[4] == -1


def test_complex_abs_is_the_scaled_form(oracle):
    """Complex.abs() of commons-math3 (SS:80): |a| sqrt(1 + (b/a)^2) with a the larger component; its special cases."""
    assert oracle.complex_abs(3.0, 4.0) == 4.0 * np.sqrt(1 + (3.0 / 4.0) ** 2)
    assert oracle.complex_abs(-4.0, 3.0) == 4.0 * np.sqrt(1 + (3.0 / -4.0) ** 2)
    assert oracle.complex_abs(0.0, 0.0) == 0.0 and oracle.complex_abs(0.0, -2.5) == 2.5 and oracle.complex_abs(-2.5, 0.0) == 2.5
    assert np.isnan(oracle.complex_abs(np.nan, 1.0)) and np.isnan(oracle.complex_abs(np.inf, np.nan))
    assert oracle.complex_abs(-np.inf, 1.0) == np.inf and oracle.complex_abs(1.0, np.inf) == np.inf
    assert oracle.complex_abs(1e300, 1e300) == 1e300 * np.sqrt(2.0)      # no overflow: the point of the scaled form
    rng = np.random.default_rng(0)
    z = rng.normal(size=(1000, 2))
    got = np.array([oracle.complex_abs(a, b) for a, b in z])
    assert np.abs(got - np.hypot(z[:, 0], z[:, 1])).max() <= 4e-16 * np.abs(got).max()   # within rounding of hypot, not equal


def test_fft_rejects_non_power_of_two(oracle):
    with pytest.raises(ValueError):
        oracle.fft_forward(np.zeros(12, complex))
    with pytest.raises(ValueError):
        oracle.compute_magnitudes(np.zeros(8 * 100, np.uint8), 0, 100, "cf32_le")
    with pytest.raises(IndexError):
        oracle.compute_magnitudes(np.zeros(8 * 64, np.uint8), 8, 64, "cf32_le")


def test_bytes_per_sample_table(oracle):
    # Global.java:67-79 incl. the fallback
    assert [oracle.bytes_per_sample(d) for d in ("cf32_le", "ci16_be", "cu8", "ci8", "cf64_le", "weird")] == [8, 4, 2, 2, 16, 8]


def test_display_conversion(oracle):
    # MC:1273-1274
    assert oracle.display_conversion(1e6, 1024) == pytest.approx(10 * np.log10(1e6 / 1024) + 20 * np.log10(1024))


@pytest.mark.parametrize("scaling", ["density", "spectrum"])
@pytest.mark.parametrize("window", ["hann", "boxcar"])
def test_welch_matches_scipy(oracle, scaling, window):
    nfft, hop, n_seg, fs = 1024, 256, 9, 2.5e6
    buf = oracle.synth_iq("cf32_le", 4, 0, (n_seg - 1) * hop + nfft)
    x = oracle.np_decode(buf, 0, (n_seg - 1) * hop + nfft, "cf32_le")
    f_ref, p_ref = scipy.signal.welch(x, fs, window=window, nperseg=nfft, noverlap=nfft - hop, nfft=nfft,
                                      detrend=False, return_onesided=False, scaling=scaling)
    f, p = oracle.welch_psd(buf, 0, "cf32_le", nfft, hop, n_seg,
                            oracle.WIN_HANN if window == "hann" else oracle.WIN_RECT,
                            oracle.PSD_DENSITY if scaling == "density" else oracle.PSD_SPECTRUM, fs)
    assert np.allclose(f, np.fft.fftshift(f_ref))
    assert np.abs(p - np.fft.fftshift(p_ref)).max() <= 1e-12 * p_ref.max()


def test_golden_fixtures(oracle):
    """tests/golden/*.npz were written by tests/golden/make_golden.py from the C oracle after
    the numpy cross-check; they freeze the expected lines so a later oracle edit cannot drift."""
    files = sorted(f for f in os.listdir(GOLDEN) if f.startswith("wf_") and f.endswith(".npz"))
    assert files, "no golden fixtures committed"
    for f in files:
        g = np.load(os.path.join(GOLDEN, f))
        dt, nfft, hop, window = str(g["datatype"]), int(g["nfft"]), int(g["hop"]), int(g["window"])
        out = oracle.waterfall(g["iq"], 0, dt, nfft, hop, g["db"].shape[0], window)
        # round 3: the fixtures were regenerated from the published commons-math3 transform (FFT_CM3); they must now be
        # reproduced to rounding of the dB -> linear conversion, and the yardstick transform must stay within the
        # reference transform's own error of them
        lin, lin_ref = 10 ** (out / 20), 10 ** (g["db"].astype(np.float64) / 20)
        assert np.abs(lin - lin_ref).max() <= 1e-13 * lin_ref.max(), f
        ex = oracle.waterfall(g["iq"], 0, dt, nfft, hop, g["db"].shape[0], window, fft=oracle.FFT_EXACT)
        assert np.abs(10 ** (ex / 20) - lin_ref).max() <= max(2e-14, cm3_lin_bound(nfft)) * lin_ref.max(), f


def test_synth_is_counter_based(oracle):
    a = oracle.synth_iq("ci16_le", 77, 0, 4096)
    b = oracle.synth_iq("ci16_le", 77, 1000, 1000)
    assert np.array_equal(a[4000:8000], b)      # any shard regenerates its own span


def test_render_known_colours(oracle):
    # MC:926-957 at hand-computable points: width 1, nfft 4, height 4 -> bins 0,1,2,3 bottom to top
    fs, nfft = 4.0, 4
    conv = oracle.display_conversion(fs, nfft)          # 10log10(1) + 20log10(4)
    db = np.array([[-100.0, -50.0, 0.0, 50.0]]) + conv  # normalised 0, 0.5, 1, clamp(1.5)
    g = oracle.render_spectrogram(db, 4, fs, -100.0, 0.0, 0)
    assert g.shape == (4, 1, 4)
    assert g[3, 0].tolist() == [0, 0, 0, 255]           # bottom row = bin 0 = black
    assert g[2, 0].tolist() == [128, 128, 128, 255]     # 0.5 * 255 = 127.5 -> Math.round -> 128
    assert g[1, 0].tolist() == [255, 255, 255, 255] and g[0, 0].tolist() == [255, 255, 255, 255]
    h = oracle.render_spectrogram(db, 4, fs, -100.0, 0.0, 1)
    assert h[3, 0].tolist() == [0, 0, 0, 255]           # n < 0.2 -> BLACK
    assert h[2, 0].tolist() == [0, 0, 255, 255]         # n = 0.5 -> RED.interpolate(YELLOW, 0) = RED  (B,G,R,A)
    assert h[1, 0].tolist() == [0, 255, 255, 255]       # n = 1 -> YELLOW
    mid = oracle.render_spectrogram(np.array([[-65.0 + conv] * 4]), 4, fs, -100.0, 0.0, 1)   # n = 0.35: BLUE->RED halfway
    assert mid[0, 0].tolist() == [128, 0, 128, 255]


def test_render_decimation_matches_numpy(oracle):
    rng = np.random.default_rng(3)
    W, N, H, fs = 37, 1024, 301, 2.5e6
    wf = rng.uniform(-160, 40, size=(W, N))
    img = oracle.render_spectrogram(wf, H, fs, -120.0, -20.0, 0)
    bins = (np.arange(H, dtype=np.float64) / H * N).astype(np.int64)            # MC:1280
    n = np.clip((wf[:, bins] - oracle.display_conversion(fs, N) + 120.0) / 100.0, 0, 1)
    ref = np.floor(n.astype(np.float32).astype(np.float64) * 255.0 + 0.5).astype(np.uint8)   # [W, H]
    assert np.array_equal(img[::-1, :, 0].T, ref) and np.all(img[..., 3] == 255)


def _cm3_transform_py(re, im):
    """A second, independent transcription of commons-math3 3.6.1's FastFourierTransformer.transformInPlace (FORWARD,
    STANDARD) in plain Python floats (IEEE doubles, no fused multiply-add) -- same published steps as
    oracle/spec_oracle.c::cm3_transform_in_place, written separately, so that a slip of the pen in either shows up as a
    bit difference."""
    import math
    n = len(re)
    if n == 1:
        return
    if n == 2:
        re[0], re[1] = re[0] + re[1], re[0] - re[1]
        im[0], im[1] = im[0] + im[1], im[0] - im[1]
        return
    j = 0                                                   # bitReversalShuffle2
    for i in range(n):
        if i < j:
            re[i], re[j] = re[j], re[i]
            im[i], im[j] = im[j], im[i]
        k = n >> 1
        while k <= j and k > 0:
            j -= k
            k >>= 1
        j += k
    for i0 in range(0, n, 4):                               # 4-term DFT
        i1, i2, i3 = i0 + 1, i0 + 2, i0 + 3
        r0, s0, r1, s1, r2, s2, r3, s3 = re[i0], im[i0], re[i2], im[i2], re[i1], im[i1], re[i3], im[i3]
        re[i0] = r0 + r1 + r2 + r3
        im[i0] = s0 + s1 + s2 + s3
        re[i1] = r0 - r2 + (s1 - s3)
        im[i1] = s0 - s2 + (r3 - r1)
        re[i2] = r0 - r1 + r2 - r3
        im[i2] = s0 - s1 + s2 - s3
        re[i3] = r0 - r2 + (s3 - s1)
        im[i3] = s0 - s2 + (r1 - r3)
    last_n0, last_log = 4, 2
    while last_n0 < n:
        n0, log_n0 = last_n0 << 1, last_log + 1
        a = 2.0 * math.pi / 2.0 ** log_n0                   # the library's W_SUB_N tables: cos / -sin at this double
        wr, wi = math.cos(a), -math.sin(a)
        for even in range(0, n, n0):
            odd = even + last_n0
            cr, ci = 1.0, 0.0
            for r in range(last_n0):
                gr, gi, hr, hi = re[even + r], im[even + r], re[odd + r], im[odd + r]
                re[even + r] = gr + cr * hr - ci * hi
                im[even + r] = gi + cr * hi + ci * hr
                re[odd + r] = gr - (cr * hr - ci * hi)
                im[odd + r] = gi - (cr * hi + ci * hr)
                cr, ci = cr * wr - ci * wi, cr * wi + ci * wr
        last_n0, last_log = n0, log_n0


@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 64, 256, 1024, 4096])
def test_cm3_c_and_python_transcriptions_agree_bit_for_bit(oracle, n):
    rng = np.random.default_rng(n)
    x = rng.normal(size=n) + 1j * rng.normal(size=n)
    re, im = list(x.real), list(x.imag)
    _cm3_transform_py(re, im)
    got = oracle.fft_forward(x, oracle.FFT_CM3)
    assert np.array_equal(got.real, np.array(re)) and np.array_equal(got.imag, np.array(im))
    # Complex.abs() the same way
    for a, b in zip(re[:32], im[:32]):
        if abs(a) < abs(b):
            q = a / b
            ref = abs(b) * np.sqrt(1 + q * q) if b != 0.0 else abs(a)
        else:
            q = b / a if a != 0.0 else 0.0
            ref = abs(a) * np.sqrt(1 + q * q) if a != 0.0 else abs(b)
        assert oracle.complex_abs(a, b) == ref
