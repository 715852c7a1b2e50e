"""CPU tests of the oracle's burst chain (no GPU): the reader of
ExtractDownConvertService.java:60-97, the build-defined down-converter, and the two traces of
AnalysisDialogController.java:219-284, each against a plain numpy restatement of the cited
Java loop and against the committed fixtures."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DTYPES = ["cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8", "cf64_le", "cf64_be"]


def np_ema(x, alpha):
    v = np.empty_like(x)
    for i in range(len(x)):
        v[i] = x[i] if i == 0 else alpha * x[i] + (1 - alpha) * v[i - 1]
    return v


@pytest.mark.parametrize("datatype", DTYPES)
def test_reader_is_bit_exact_with_the_decode_table(oracle, datatype):
    iq = oracle.synth_iq(datatype, 5, 0, 700)
    re, im = oracle.extract_iq(iq, 13, 600, datatype)
    z = oracle.np_decode(iq, 13 * oracle.bytes_per_sample(datatype), 600, datatype)
    assert np.array_equal(re, z.real) and np.array_equal(im, z.imag)


def test_reader_quirks(oracle):
    # EDC:94-96: a datatype the reader does not know is read as float pairs (no zero branch)
    x = np.arange(40, dtype="<f4")
    re, im = oracle.extract_iq(x.view(np.uint8), 2, 10, "xx99_le")
    assert np.array_equal(re, x[4:24:2]) and np.array_equal(im, x[5:24:2])
    # EDC:60-67: no cf64 case in the stride table -> 8 bytes: Q of sample i is I of sample i + 1
    d = np.arange(40, dtype="<f8")
    re, im = oracle.extract_iq(d.view(np.uint8), 3, 10, "cf64_le", ref_cf64_stride8=True)
    assert np.array_equal(re, d[3:13]) and np.array_equal(im, d[4:14])
    re, im = oracle.extract_iq(d.view(np.uint8), 3, 10, "cf64_le")
    assert np.array_equal(re, d[6:26:2]) and np.array_equal(im, d[7:26:2])
    # the buffer getters throw past the end
    with pytest.raises(IndexError):
        oracle.extract_iq(d.view(np.uint8), 11, 10, "cf64_le")
    oracle.extract_iq(d.view(np.uint8), 10, 10, "cf64_le")


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("down", [1, 4, 10])
def test_down_converter_specification(oracle, mode, down):
    rng = np.random.default_rng(down + mode)
    n, f_off = 1200, 0.0831
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    dr, di = oracle.down_convert(x.real, x.imag, f_off, down, mode)
    h, c = oracle.down_convert_taps(down, mode)
    assert len(h) == (down if mode == 0 else 8 * down + 1) and abs(h.sum() - 1) < 1e-14
    assert len(dr) == n // down
    t = f_off * np.arange(n)
    xm = x * np.exp(-2j * np.pi * (t - np.floor(t)))
    full = np.convolve(xm, h)  # full[j] = sum_k h[k] xm[j - k]
    chk = full[np.arange(n // down) * down + c]
    assert np.abs((dr + 1j * di) - chk).max() <= 1e-13
    if down == 1 and mode == 1:  # cut-off 0.5: the sinc collapses to a delta
        assert np.abs((dr + 1j * di) - xm).max() <= 1e-13


def test_down_converter_moves_a_tone_to_dc(oracle):
    n, f0, down = 8000, 0.21, 8
    x = np.exp(2j * np.pi * f0 * np.arange(n))
    for mode in (0, 1):
        dr, di = oracle.down_convert(x.real, x.imag, f0, down, mode)
        y = (dr + 1j * di)[8:-8]
        assert np.abs(y - 1).max() < 1e-9   # unit DC gain, tone now at 0 Hz
    # a tone half-way to the first boxcar null is attenuated by the low-pass far more than by the boxcar
    dr0, di0 = oracle.down_convert(x.real, x.imag, f0 - 0.75 / down, down, 0)
    dr1, di1 = oracle.down_convert(x.real, x.imag, f0 - 0.75 / down, down, 1)
    assert np.abs(dr1 + 1j * di1)[8:-8].max() < 0.2 * np.abs(dr0 + 1j * di0)[8:-8].max()


@pytest.mark.parametrize("alpha", [0.0, 0.05, 0.5, 1.0])
def test_traces_match_the_java_loops(oracle, alpha):
    rng = np.random.default_rng(3)
    n, fs, fc = 3000, 48e3, 1e6
    x = np.exp(2j * np.pi * 0.11 * np.arange(n)) * (1 + 0.3 * rng.standard_normal(n)) + 0.05 * (
        rng.standard_normal(n) + 1j * rng.standard_normal(n))
    mag = oracle.magnitude_trace(x.real, x.imag, alpha)
    assert np.abs(mag - 20 * np.log10(np_ema(np.abs(x), alpha))).max() <= 1e-11
    frq = oracle.inst_freq_trace(x.real, x.imag, alpha, fs, fc)
    ph = np.arctan2(x.imag, x.real)
    d = ph[1:] - ph[:-1]
    d = np.where(d > np.pi, d - 2 * np.pi, np.where(d < -np.pi, d + 2 * np.pi, d))
    assert frq.shape == (n - 1,)
    assert np.abs(frq - (np_ema(d / (2 * np.pi) * fs, alpha) + fc)).max() <= 1e-9 * fc
    if alpha == 1.0:
        assert abs(np.median(frq) - (fc + 0.11 * fs)) < 0.02 * fs


def test_trace_edge_cases(oracle):
    z = np.zeros(4)
    assert np.all(np.isneginf(oracle.magnitude_trace(z, z, 0.3)))        # log10(0): dropped by ADC:239-242
    assert oracle.inst_freq_trace(z[:1], z[:1], 0.3, 1.0).shape == (0,)
    assert oracle.magnitude_trace(z[:0], z[:0], 0.3).shape == (0,)
    # phase wrap: a step of +0.75 cycles is read as -0.25 cycles (ADC:270-275)
    x = np.exp(2j * np.pi * 0.75 * np.arange(6))
    f = oracle.inst_freq_trace(x.real, x.imag, 1.0, 1.0)
    assert np.allclose(f, -0.25)


def test_burst_fixtures(oracle):
    files = sorted(f for f in os.listdir(GOLDEN) if f.startswith("burst_"))
    assert files, "no burst fixtures committed"
    for f in files:
        g = np.load(os.path.join(GOLDEN, f))
        dt = str(g["datatype"])
        re, im = oracle.extract_iq(g["iq"], int(g["start"]), int(g["count"]), dt)
        assert np.array_equal(re, g["re"]) and np.array_equal(im, g["im"]), f
        for mode in (0, 1):
            dr, di = oracle.down_convert(re, im, float(g["freq_off"]), int(g["down"]), mode)
            assert np.abs(dr - g["dc%d_re" % mode]).max() <= 1e-14 and np.abs(di - g["dc%d_im" % mode]).max() <= 1e-14, f
        assert np.abs(oracle.magnitude_trace(g["dc0_re"], g["dc0_im"], float(g["alpha"])) - g["mag"]).max() <= 1e-12
        assert np.abs(oracle.inst_freq_trace(g["dc0_re"], g["dc0_im"], float(g["alpha"]), float(g["fs"]),
                                             float(g["center"])) - g["freq"]).max() <= 1e-6
