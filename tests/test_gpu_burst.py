"""GPU parity of the burst chain through the C ABI:
spec_extract_iq / spec_down_convert (ExtractDownConvertService.java:54-117) and
spec_magnitude_trace / spec_inst_freq_trace (AnalysisDialogController.java:219-284).

Tolerances (fp64 pipeline):
  reader            bit-exact (the conversions are exact in fp64)
  down-converter    |dy| <= 1e-12 * max|x|   (device sin/cos vs libm, FMA contraction)
  magnitude trace   |d dB| <= 1e-9 where the smoothed magnitude is >= 1e-9 of its maximum
  frequency trace   |d Hz| <= 1e-9 * fs      (atan2 of the device library vs libm; scan order)
"""
import os

import numpy as np
import pytest

import spectral_analyzer_amd as sa

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DTYPES = ["cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8", "cf64_le", "cf64_be"]


@pytest.fixture(scope="module")
def edc(svc):
    return sa.ExtractDownConvertService(svc)


def check_mag(got, ref):
    v = 10 ** (ref / 20)
    ok = v >= 1e-9 * v.max()
    assert np.abs(got[ok] - ref[ok]).max() <= 1e-9


@pytest.mark.parametrize("datatype", DTYPES)
def test_reader_bit_exact(edc, oracle, datatype):
    iq = oracle.synth_iq(datatype, 41, 5, 5000)
    ref = oracle.extract_iq(iq, 123, 4000, datatype)
    got = edc.extract_iq(iq, 123, 4000, datatype)
    assert got.dtype == np.float64 and got.shape == (2, 4000)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


def test_reader_quirks_and_errors(svc, edc, oracle):
    x = np.arange(40, dtype="<f4")
    got = edc.extract_iq(x.view(np.uint8), 2, 10, "xx99_le")           # EDC:94-96
    assert np.array_equal(got[0], x[4:24:2]) and np.array_equal(got[1], x[5:24:2])
    d = np.arange(40, dtype="<f8")
    got = edc.extract_iq(d.view(np.uint8), 3, 10, "cf64_le")
    assert np.array_equal(got[0], d[6:26:2])
    with sa.SpectralService(0, ref_edc_cf64_stride8=True) as s8:            # EDC:60-67
        got = sa.ExtractDownConvertService(s8).extract_iq(d.view(np.uint8), 3, 10, "cf64_le")
        assert np.array_equal(got[0], d[3:13]) and np.array_equal(got[1], d[4:14])
    with pytest.raises(IndexError):
        edc.extract_iq(d.view(np.uint8), 11, 10, "cf64_le")
    assert edc.extract_iq(d.view(np.uint8), 10, 10, "cf64_le").shape == (2, 10)
    assert edc.extract_iq(d.view(np.uint8), 0, 0, "cf64_le").shape == (2, 0)
    with pytest.raises(ValueError):
        edc.extract_and_down_convert(d.view(np.uint8), 0, 10, "cf64_le", 0.1, 0, True)


@pytest.mark.parametrize("datatype", ["cf32_le", "ci16_be", "cu8", "cf64_le"])
@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("down", [1, 5, 16])
def test_down_converter_parity(edc, oracle, datatype, fast, down):
    n, f_off = 20000, -0.1234
    iq = oracle.synth_iq(datatype, 43, 0, n + 100)
    re, im = oracle.extract_iq(iq, 50, n, datatype)
    ref = oracle.down_convert(re, im, f_off, down, 0 if fast else 1)
    got = edc.extract_and_down_convert(iq, 50, n, datatype, f_off, down, fast)
    assert got.shape == (2, n // down)
    m = max(np.abs(re).max(), np.abs(im).max())
    assert np.abs(got[0] - ref[0]).max() <= 1e-12 * m and np.abs(got[1] - ref[1]).max() <= 1e-12 * m


@pytest.mark.parametrize("n", [1, 2, 7, 2048, 2049, 100003, 1 << 21])
@pytest.mark.parametrize("alpha", [0.0, 0.02, 0.7, 1.0])
def test_traces_parity(svc, oracle, n, alpha):
    rng = np.random.default_rng(n)
    x = np.exp(2j * np.pi * 0.07 * np.arange(n)) * (1 + 0.2 * rng.standard_normal(n)) + 0.02 * (
        rng.standard_normal(n) + 1j * rng.standard_normal(n))
    data = np.stack([x.real, x.imag])
    fs, fc = 250e3, 433.92e6
    check_mag(svc.magnitude_trace(data, alpha), oracle.magnitude_trace(data[0], data[1], alpha))
    got = svc.inst_freq_trace(data, alpha, fs, fc)
    ref = oracle.inst_freq_trace(data[0], data[1], alpha, fs, fc)
    assert got.shape == (n - 1,)
    if n > 1:
        assert np.abs(got - ref).max() <= 1e-9 * fs


def test_traces_edge_cases(svc):
    z = np.zeros((2, 5))
    assert np.all(np.isneginf(svc.magnitude_trace(z, 0.3)))
    assert svc.magnitude_trace(z[:, :0], 0.3).shape == (0,)
    assert svc.inst_freq_trace(z[:, :1], 0.3, 1.0).shape == (0,)
    x = np.exp(2j * np.pi * 0.75 * np.arange(6))
    assert np.allclose(svc.inst_freq_trace(np.stack([x.real, x.imag]), 1.0, 1.0), -0.25)   # ADC:270-275
    with pytest.raises(ValueError):
        svc.magnitude_trace(np.zeros((3, 4)), 0.5)


def test_device_resident_chain(svc, edc, oracle):
    """reader -> down-converter -> traces without leaving the GPU (torch tensors in and out)."""
    import torch
    n, down, f_off, alpha = 1 << 20, 8, 0.05, 0.1
    iq = svc.synth_iq("ci16_le", 9, 0, n)
    host = iq.cpu().numpy()
    d = edc.extract_and_down_convert(iq, 0, n, "ci16_le", f_off, down, True)
    assert d.is_cuda and d.shape == (2, n // down) and d.dtype == torch.float64
    mag, frq = svc.magnitude_trace(d, alpha), svc.inst_freq_trace(d, alpha, 1e6 / down, 0.0)
    torch.cuda.synchronize()
    re, im = oracle.extract_iq(host, 0, n, "ci16_le")
    rr, ri = oracle.down_convert(re, im, f_off, down, 0)
    assert np.abs(d[0].cpu().numpy() - rr).max() <= 1e-12 and np.abs(d[1].cpu().numpy() - ri).max() <= 1e-12
    check_mag(mag.cpu().numpy(), oracle.magnitude_trace(rr, ri, alpha))
    assert np.abs(frq.cpu().numpy() - oracle.inst_freq_trace(rr, ri, alpha, 1e6 / down, 0.0)).max() <= 1e-9 * 1e6 / down


def test_burst_fixtures(svc, edc):
    for f in sorted(f for f in os.listdir(GOLDEN) if f.startswith("burst_")):
        g = np.load(os.path.join(GOLDEN, f))
        dt, start, count, down = str(g["datatype"]), int(g["start"]), int(g["count"]), int(g["down"])
        got = edc.extract_iq(g["iq"], start, count, dt)
        assert np.array_equal(got[0], g["re"]) and np.array_equal(got[1], g["im"]), f
        for mode in (0, 1):
            y = edc.extract_and_down_convert(g["iq"], start, count, dt, float(g["freq_off"]), down, mode == 0)
            assert np.abs(y[0] - g["dc%d_re" % mode]).max() <= 1e-12 and np.abs(y[1] - g["dc%d_im" % mode]).max() <= 1e-12, f
        data = np.stack([g["dc0_re"], g["dc0_im"]])
        check_mag(svc.magnitude_trace(data, float(g["alpha"])), g["mag"])
        assert np.abs(svc.inst_freq_trace(data, float(g["alpha"]), float(g["fs"]), float(g["center"])) - g["freq"]).max() \
            <= 1e-9 * float(g["fs"])


def test_psd_of_a_device_resident_burst(svc, edc, oracle):
    """The PSD dialog's call (ADC:308-312) on the down-converter's output without leaving the GPU equals the
    same call on a host copy, and the build-defined Welch of the oracle."""
    n, down = 1 << 18, 4
    iq = svc.synth_iq("cf32_le", 17, 0, n)
    d = edc.extract_and_down_convert(iq, 0, n, "cf32_le", 0.11, down, False)
    on_dev = svc.calculate_psd_welch(d, 2.5e5, 8192)
    host = d.cpu().numpy()
    on_host = svc.calculate_psd_welch(host, 2.5e5, 8192)
    assert np.array_equal(on_dev, on_host)
    raw = np.empty(2 * host.shape[1]); raw[0::2], raw[1::2] = host[0], host[1]
    f, p = oracle.welch_psd(raw.view(np.uint8), 0, "cf64_le", 8192, 4096, (host.shape[1] - 8192) // 4096 + 1, fs=2.5e5)
    # the dialog's row is dB (ADC:319-328, 612, 675, 751): 2e-6 relative in power = 8.7e-6 dB
    assert np.allclose(on_dev[0], f) and np.abs(on_dev[1] - 10 * np.log10(p + 1e-20)).max() <= 1e-5


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_random_burst_requests_match_oracle(svc, edc, oracle, seed):
    """The whole chain on random requests: datatype, start, count, frequency offset, decimation (both filters, including
    factors past the one-pass FIR kernel's span) and smoothing factor -- reader bit for bit, the rest to the tolerances above."""
    rng = np.random.default_rng(seed)
    for _ in range(25):
        dt = str(rng.choice(DTYPES))
        count = int(rng.choice([1, 2, 3, int(rng.integers(4, 300)), int(rng.integers(300, 60000))]))
        start = int(rng.integers(0, 100))
        down = int(rng.choice([1, 2, 3, int(rng.integers(4, 40)), int(rng.integers(40, 320))]))
        fast = bool(rng.integers(0, 2))
        f_off = float(rng.uniform(-0.5, 0.5))
        alpha = float(rng.choice([0.0, 1.0, rng.uniform(0.001, 0.999)]))
        tag = (dt, count, start, down, fast, f_off, alpha)
        iq = oracle.synth_iq(dt, int(rng.integers(1, 1 << 30)), 0, start + count + 3)
        re, im = oracle.extract_iq(iq, start, count, dt)
        got = edc.extract_iq(iq, start, count, dt)
        assert np.array_equal(got[0], re) and np.array_equal(got[1], im), tag
        ref = oracle.down_convert(re, im, f_off, down, 0 if fast else 1)
        y = edc.extract_and_down_convert(iq, start, count, dt, f_off, down, fast)
        assert y.shape == (2, count // down), tag
        if y.shape[1] == 0:
            continue
        m = max(np.abs(re).max(), np.abs(im).max(), 1e-300)
        assert np.abs(y[0] - ref[0]).max() <= 1e-12 * m and np.abs(y[1] - ref[1]).max() <= 1e-12 * m, tag
        data = np.stack([ref[0], ref[1]])
        mag, mref = svc.magnitude_trace(data, alpha), oracle.magnitude_trace(ref[0], ref[1], alpha)
        v = 10 ** (mref / 20)
        ok = v >= 1e-9 * max(v.max(), 1e-300)
        assert ok.sum() == 0 or np.abs(mag[ok] - mref[ok]).max() <= 1e-9, tag
        fs = 1e6 / down
        frq = svc.inst_freq_trace(data, alpha, fs, 1e8)
        assert frq.shape == (data.shape[1] - 1,), tag
        if frq.size:
            assert np.abs(frq - oracle.inst_freq_trace(ref[0], ref[1], alpha, fs, 1e8)).max() <= 1e-9 * fs, tag
