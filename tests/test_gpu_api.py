"""GPU tests of the drop-in boundary: the exact computeMagnitudes call, its error
behaviour, the batched line loop with EOF fill, fp64 strict parity, Welch PSD, device-
resident buffers, the committed golden fixtures, and full-size property checks."""
import os

import numpy as np
import pytest

import spectral_analyzer_amd as sa
from test_gpu_parity import check_fp32, check_fp64, fp64_pow_tol

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DTYPES = ["cf32_le", "cf32_be", "ci16_le", "ci16_be", "cu8", "ci8", "cf64_le", "cf64_be"]


# ---- SpectralService.computeMagnitudes (SS:33-85), fp64 pipeline -----------------------
@pytest.mark.parametrize("datatype", DTYPES)
@pytest.mark.parametrize("nfft", [64, 1024, 8192])
def test_compute_magnitudes_fp64_parity(svc, oracle, datatype, nfft):
    iq = oracle.synth_iq(datatype, 31, 17, nfft + 50)
    start = 3 * oracle.bytes_per_sample(datatype)
    ref = oracle.compute_magnitudes(iq, start, nfft, datatype, cf64_decode=True)
    got = svc.compute_magnitudes(iq, start, nfft, datatype)
    assert got.dtype == np.float64 and got.shape == (nfft,)
    check_fp64(got[None, :], ref[None, :])


def test_compute_magnitudes_known_answers(svc):
    n = 256
    assert np.all(svc.compute_magnitudes(np.zeros(8 * n, np.uint8), 0, n, "cf32_le") == -200.0)       # K1
    assert np.all(svc.compute_magnitudes(np.ones(8 * n, np.uint8), 0, n, "what") == -200.0)           # K2
    x = np.zeros(2 * n, "<f4"); x[0] = 1.0
    assert np.allclose(svc.compute_magnitudes(x, 0, n, "cf32_le"), 20 * np.log10(1 + 1e-10), atol=1e-9)  # K3
    dc = np.zeros(2 * n, "<f4"); dc[0::2] = 1.0
    out = svc.compute_magnitudes(dc, 0, n, "cf32_le")
    assert out[n // 2] == pytest.approx(20 * np.log10(n), abs=1e-9) and np.delete(out, n // 2).max() < -150   # K4
    k = 37
    tone = np.exp(2j * np.pi * k * np.arange(n) / n)
    t = np.empty(2 * n, "<f8"); t[0::2], t[1::2] = tone.real, tone.imag
    out = svc.compute_magnitudes(t, 0, n, "cf64_le")
    assert int(np.argmax(out)) == (k + n // 2) % n and out.max() == pytest.approx(20 * np.log10(n), abs=1e-9)  # K5


def test_compute_magnitudes_byte_order_is_the_buffers(svc, oracle):
    le = oracle.synth_iq("ci16_le", 8, 0, 512)
    be = oracle.synth_iq("ci16_be", 8, 0, 512)
    a = svc.compute_magnitudes(le, 0, 512, "ci16_le")
    b = svc.compute_magnitudes(be, 0, 512, "ci16_be")
    c = svc.compute_magnitudes(be, 0, 512, "ci16", big_endian=True)     # startsWith match, explicit order
    assert np.array_equal(a, b) and np.array_equal(a, c)                 # K7


def test_compute_magnitudes_errors(svc):
    buf = np.zeros(8 * 64, np.uint8)
    with pytest.raises(ValueError):            # commons-math3: not a power of two
        svc.compute_magnitudes(buf, 0, 48, "cf32_le")
    with pytest.raises(ValueError):
        svc.compute_magnitudes(buf, 0, 0, "cf32_le")
    with pytest.raises(IndexError):            # ByteBuffer getters: IndexOutOfBoundsException
        svc.compute_magnitudes(buf, 8, 64, "cf32_le")
    with pytest.raises(IndexError):
        svc.compute_magnitudes(buf, -8, 64, "cf32_le")
    # an unknown datatype never touches the buffer (SS:60-63), so no range error
    assert np.all(svc.compute_magnitudes(buf, 10 ** 6, 64, "nope") == -200.0)


def test_cf64_reference_defect_flag(oracle):
    iq = oracle.synth_iq("cf64_le", 1, 0, 256)
    with sa.SpectralService(0, ref_cf64_zero=True) as s:
        assert np.all(s.compute_magnitudes(iq, 0, 256, "cf64_le") == -200.0)     # SS:35-63 has no cf64 branch
        assert np.all(s.compute_waterfall(iq, 0, 256, "cf64_le", 1) == -200.0)
    with sa.SpectralService(0) as s:
        assert s.compute_magnitudes(iq, 0, 256, "cf64_le").max() > 0            # EDC:79-81 decode


# ---- the line loop (MC:980-999) ----------------------------------------------------------
@pytest.mark.parametrize("datatype", ["cf32_le", "ci16_be", "cu8"])
def test_eof_fill_and_start_byte(svc, oracle, datatype):
    nfft = 256
    bps = oracle.bytes_per_sample(datatype)
    iq = oracle.synth_iq(datatype, 6, 0, 1000)
    for start in (0, 300 * bps):
        ref = oracle.waterfall(iq, start, datatype, nfft, nfft, 6)      # reference hop == nfft
        got = svc.compute_waterfall(iq, start, nfft, datatype, 6)       # hop defaults to nfft
        valid = oracle.count_lines(iq.size, start, datatype, nfft, nfft)
        assert svc.count_lines(iq.size, start, datatype, nfft, nfft) == valid < 6
        assert np.all(got[valid:] == -150.0)                             # MC:994-998
        check_fp32(got[:valid], ref[:valid], nfft)
    got = svc.compute_waterfall(iq, 0, nfft, datatype, 4, eof_fill=-99.5)
    assert np.all(got[3:] == -99.5)
    assert svc.compute_waterfall(iq, 0, nfft, datatype, 0).shape == (0, nfft)
    assert np.all(svc.compute_waterfall(iq, iq.size + 10, nfft, datatype, 2) == -150.0)


@pytest.mark.parametrize("hop", [1, 100, 256, 300, 1024])
def test_any_hop(svc, oracle, hop):
    nfft, n_lines, dt = 256, 11, "ci8"
    iq = oracle.synth_iq(dt, hop, 0, (n_lines - 1) * hop + nfft)
    check_fp32(svc.compute_waterfall(iq, 0, nfft, dt, n_lines, hop=hop),
               oracle.waterfall(iq, 0, dt, nfft, hop, n_lines), nfft)


@pytest.mark.parametrize("datatype", ["cf32_le", "ci16_le", "cf64_be"])
@pytest.mark.parametrize("nfft", [128, 4096, 8192])
def test_fp64_outputs_strict(svc, oracle, datatype, nfft):
    hop, n_lines = nfft // 4, 5
    iq = oracle.synth_iq(datatype, 12, 0, (n_lines - 1) * hop + nfft)
    for window in (sa.WIN_RECT, sa.WIN_HANN):
        ref = oracle.waterfall(iq, 0, datatype, nfft, hop, n_lines, window)
        got = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop, window=window, out_fmt=sa.OUT_DB20_F64)
        assert got.dtype == np.float64
        check_fp64(got, ref)
        p_ref = oracle.waterfall(iq, 0, datatype, nfft, hop, n_lines, window, power=True)
        p = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop, window=window, out_fmt=sa.OUT_POW_F64)
        assert np.abs(p - p_ref).max() <= fp64_pow_tol(nfft) * p_ref.max()


@pytest.mark.parametrize("nfft", [512, 4096])
def test_power_output_and_parseval(svc, oracle, nfft):
    dt, n_lines = "ci16_le", 20
    iq = oracle.synth_iq(dt, 3, 0, n_lines * nfft)
    p = svc.compute_waterfall(iq, 0, nfft, dt, n_lines, out_fmt=sa.OUT_POW_F32).astype(np.float64)
    p_ref = oracle.waterfall(iq, 0, dt, nfft, nfft, n_lines, power=True)
    assert np.abs(p - p_ref).max() <= 2e-6 * p_ref.max()
    x = oracle.np_decode(iq, 0, n_lines * nfft, dt).reshape(n_lines, nfft)
    assert np.allclose(p.sum(axis=1), nfft * (np.abs(x) ** 2).sum(axis=1), rtol=2e-6)      # K9


def test_argument_errors(svc, oracle):
    iq = oracle.synth_iq("cf32_le", 1, 0, 4096)
    with pytest.raises(ValueError):
        svc.compute_waterfall(iq, 0, 1000, "cf32_le", 1)
    with pytest.raises(ValueError):
        svc.compute_waterfall(iq, 0, 1024, "cf32_le", 1, hop=0)
    with pytest.raises(ValueError):
        svc.compute_waterfall(iq, 0, 1024, "cf32_le", 1, window=7)
    with pytest.raises(ValueError):
        svc.set_option("no_such_knob", 1)


# ---- device-resident buffers (the bench path) -------------------------------------------------
def test_device_buffers_and_alignment(svc, oracle):
    import torch
    dt, nfft, hop, n_lines = "ci16_le", 1024, 512, 9
    iq = oracle.synth_iq(dt, 2, 0, (n_lines - 1) * hop + nfft + 3)
    d = torch.from_numpy(iq).cuda()
    ref = oracle.waterfall(iq, 4, dt, nfft, hop, n_lines + 1)
    got = svc.compute_waterfall(d, 4, nfft, dt, n_lines + 1, hop=hop)
    assert got.is_cuda and got.shape == (n_lines + 1, nfft)
    torch.cuda.synchronize()
    g = got.cpu().numpy()
    assert np.all(g[-1] == -150.0)
    check_fp32(g[:-1], ref[:-1], nfft)
    with pytest.raises(ValueError, match="aligned"):
        svc.compute_waterfall(d, 1, nfft, dt, 1, hop=hop)      # odd byte offset for 2-byte components
    with pytest.raises(ValueError):
        svc.compute_waterfall(d.cpu(), 0, nfft, dt, 1)          # torch CPU tensor is not a device buffer


@pytest.mark.parametrize("datatype", ["cf32_le", "ci16_le", "ci16_be", "cu8", "ci8", "cf64_le"])
def test_device_synth_matches_cpu_generator(svc, oracle, datatype):
    n, first = 5000, 123456789
    dev = svc.synth_iq(datatype, 0x5EC7A11A, first, n).cpu().numpy()
    host = oracle.synth_iq(datatype, 0x5EC7A11A, first, n)
    a = oracle.np_decode(dev, 0, n, datatype)
    b = oracle.np_decode(host, 0, n, datatype)
    tol = {"cu8": 1 / 128 + 1e-6, "ci8": 1 / 128 + 1e-6}.get(datatype, 1 / 32768 + 2e-6)   # one quantisation step
    assert np.abs(a - b).max() <= tol


# ---- Welch PSD (ADC:303-313 call site) ----------------------------------------------------------
@pytest.mark.parametrize("nfft,hop,n_seg", [(1024, 512, 7), (8192, 4096, 5), (16384, 4096, 16), (256, 64, 33)])
@pytest.mark.parametrize("datatype", ["cf32_le", "ci16_le", "cu8"])
def test_welch_matches_oracle(svc, oracle, datatype, nfft, hop, n_seg):
    fs = 2.4e6
    iq = oracle.synth_iq(datatype, 5, 0, (n_seg - 1) * hop + nfft)
    for window, scaling in ((sa.WIN_HANN, sa.PSD_DENSITY), (sa.WIN_RECT, sa.PSD_SPECTRUM)):
        f_ref, p_ref = oracle.welch_psd(iq, 0, datatype, nfft, hop, n_seg, window, scaling, fs)
        f, p = svc.welch_psd(iq, 0, datatype, fs, nfft=nfft, hop=hop, n_seg=n_seg, window=window, scaling=scaling)
        assert np.array_equal(f, f_ref) and p.shape == (1, nfft)
        assert np.abs(p[0] - p_ref).max() <= 5e-6 * p_ref.max()
        _, pdb = svc.welch_psd(iq, 0, datatype, fs, nfft=nfft, hop=hop, n_seg=n_seg, window=window, scaling=scaling, db=True)
        _, pdb_ref = oracle.welch_psd(iq, 0, datatype, nfft, hop, n_seg, window, scaling, fs, db=True)
        strong = p_ref >= 1e-3 * p_ref.max()
        assert np.abs(pdb[0] - pdb_ref)[strong].max() <= 2e-3


def test_welch_batch_and_defaults(svc, oracle):
    import torch
    dt, nfft, hop, n_seg, n_psd, fs = "cf32_le", 1024, 512, 6, 5, 1e6
    per = (n_seg - 1) * hop + nfft
    iq = oracle.synth_iq(dt, 13, 0, per * n_psd)
    f, p = svc.welch_psd(torch.from_numpy(iq).cuda(), 0, dt, fs, nfft=nfft, hop=hop, n_seg=n_seg,
                         n_psd=n_psd, psd_stride_bytes=per * 8)
    torch.cuda.synchronize()
    p = p.cpu().numpy()
    for b in range(n_psd):
        _, ref = oracle.welch_psd(iq, b * per * 8, dt, nfft, hop, n_seg, oracle.WIN_HANN, oracle.PSD_DENSITY, fs)
        assert np.abs(p[b] - ref).max() <= 5e-6 * ref.max()
    # defaults: hop = nfft/2, every whole segment (the short-signal rule of ADC:303-313 is the caller's)
    _, p_all = svc.welch_psd(iq, 0, dt, fs, nfft=nfft)
    n_all = oracle.count_lines(iq.size, 0, dt, nfft, nfft // 2)
    _, ref = oracle.welch_psd(iq, 0, dt, nfft, nfft // 2, n_all, oracle.WIN_HANN, oracle.PSD_DENSITY, fs)
    assert np.abs(p_all[0] - ref).max() <= 5e-6 * ref.max()
    with pytest.raises(IndexError):
        svc.welch_psd(iq, 0, dt, fs, nfft=nfft, hop=hop, n_seg=10 ** 6)


@pytest.mark.parametrize("datatype,nfft,hop,n_seg,window,scaling", [
    ("cf32_le", 2048, 512, 9, sa.WIN_HANN, sa.PSD_DENSITY), ("ci16_le", 4096, 2048, 5, sa.WIN_HANN, sa.PSD_DENSITY),
    ("cf32_le", 16384, 4096, 3, sa.WIN_HANN, sa.PSD_DENSITY), ("cu8", 2048, 2048, 4, sa.WIN_RECT, sa.PSD_SPECTRUM)])
def test_welch_large_batch_finished_in_kernel(svc, oracle, datatype, nfft, hop, n_seg, window, scaling):
    """>= two PSDs per CU with whole-workgroup lines: the workgroup that walked a PSD's segments finishes it (no
    slabs, no second launch).  Same numbers as the two-launch form forced by "welch_two_pass" to fp32 summation
    order, and the oracle's on sampled PSDs; linear and dB."""
    import torch
    n_psd, fs = 600, 2.5e6
    bps = sa.bytes_per_sample(datatype)
    per = (n_seg - 1) * hop + nfft
    iq = svc.synth_iq(datatype, 21, 0, per * n_psd)
    try:
        res = {}
        for two in (0, 1):
            svc.set_option("welch_two_pass", two)
            _, p = svc.welch_psd(iq, 0, datatype, fs, nfft=nfft, hop=hop, n_seg=n_seg, window=window, scaling=scaling, n_psd=n_psd,
                               psd_stride_bytes=per * bps)
            _, pdb = svc.welch_psd(iq, 0, datatype, fs, nfft=nfft, hop=hop, n_seg=n_seg, window=window, scaling=scaling,
                                   n_psd=n_psd, psd_stride_bytes=per * bps, db=True)
            torch.cuda.synchronize()
            res[two] = (p.clone(), pdb.clone())
    finally:
        svc.set_option("welch_two_pass", 0)
    p, pdb = res[0]
    assert p.shape == (n_psd, nfft)
    assert float(((p - res[1][0]).abs().amax(dim=1) / p.amax(dim=1)).max()) <= 2e-6
    assert float((pdb - res[1][1]).abs().max()) <= 1e-3
    host = iq.cpu().numpy()
    for b in (0, 299, n_psd - 1):
        _, ref = oracle.welch_psd(host, b * per * bps, datatype, nfft, hop, n_seg, window, scaling, fs)
        assert np.abs(p[b].cpu().numpy() - ref).max() <= 5e-6 * ref.max()
        _, ref_db = oracle.welch_psd(host, b * per * bps, datatype, nfft, hop, n_seg, window, scaling,
                                     fs, db=True)
        strong = ref >= 1e-6 * ref.max()
        assert np.abs(pdb[b].cpu().numpy() - ref_db)[strong].max() <= 2e-3


# ---- committed golden vectors --------------------------------------------------------------------
def test_golden_fixtures_on_gpu(svc):
    files = sorted(f for f in os.listdir(GOLDEN) if f.startswith("wf_") and f.endswith(".npz"))
    assert files
    for f in files:
        g = np.load(os.path.join(GOLDEN, f))
        dt, nfft, hop, window = str(g["datatype"]), int(g["nfft"]), int(g["hop"]), int(g["window"])
        ref = g["db"]
        got = svc.compute_waterfall(g["iq"], 0, nfft, dt, ref.shape[0], hop=hop, window=window)
        assert np.all(got[-1] == -150.0), f
        check_fp32(got[:-1], ref[:-1], nfft)
        got64 = svc.compute_waterfall(g["iq"], 0, nfft, dt, ref.shape[0], hop=hop, window=window, out_fmt=sa.OUT_DB20_F64)
        check_fp64(got64[:-1], ref[:-1])


# ---- full-size property checks (BASELINE configs 2 and 3 sizes per GPU) -------------------------
@pytest.mark.parametrize("datatype", ["cf32_le", "ci16_le"])
def test_full_size_properties(svc, oracle, datatype):
    import torch
    nfft, hop, S = 4096, 2048, 1 << 30
    bps = sa.bytes_per_sample(datatype)
    n_lines = (S - nfft) // hop + 1
    iq = svc.synth_iq(datatype, 0x5EC7A11A, 0, S)
    out = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines + 2, hop=hop)
    torch.cuda.synchronize()
    assert out.shape == (n_lines + 2, nfft)
    assert bool((out[n_lines:] == -150.0).all())                       # EOF lines
    assert bool(torch.isfinite(out).all())
    # the 0.123 cycles/sample tone peaks at the same bin of every line
    peak = out[:n_lines].argmax(dim=1)
    k = int(round(0.123 * nfft)) + nfft // 2
    assert bool(((peak - k).abs() <= 1).all())
    # sampled lines against the oracle on the very bytes the GPU read
    rng = np.random.default_rng(1)
    for ln in [0, 1, n_lines - 1] + [int(x) for x in rng.integers(0, n_lines, 5)]:
        raw = iq[ln * hop * bps:(ln * hop + nfft) * bps].cpu().numpy()
        ref = oracle.waterfall(raw, 0, datatype, nfft, hop, 1)
        check_fp32(out[ln].cpu().numpy()[None, :], ref, nfft)
    # Parseval on a block of lines (power output of the same path)
    blk = 64
    p = svc.compute_waterfall(iq, 0, nfft, datatype, blk, hop=hop, out_fmt=sa.OUT_POW_F32)
    torch.cuda.synchronize()
    x = oracle.np_decode(iq[:((blk - 1) * hop + nfft) * bps].cpu().numpy(), 0, (blk - 1) * hop + nfft, datatype)
    for ln in (0, 17, blk - 1):
        e = (np.abs(x[ln * hop:ln * hop + nfft]) ** 2).sum() * nfft
        assert float(p[ln].double().sum()) == pytest.approx(e, rel=5e-6)
    del out, iq, p
    torch.cuda.empty_cache()


# ---- lines longer than the LDS: four-step path (spec_k_large.hip) ----------------------------------
@pytest.mark.parametrize("datatype,nfft,fmt", [("cf32_le", 32768, "f32"), ("ci16_le", 65536, "f32"),
                                               ("cf64_le", 65536, "f64"), ("cf64_be", 16384, "f64"),
                                               ("cu8", 32768, "f64"), ("cf32_le", 65536, "f64")])
def test_large_nfft(svc, oracle, datatype, nfft, fmt):
    hop, n_lines = nfft // 2, 5
    iq = oracle.synth_iq(datatype, 21, 3, (n_lines - 1) * hop + nfft)
    for window in (sa.WIN_RECT, sa.WIN_HANN):
        ref = oracle.waterfall(iq, 0, datatype, nfft, hop, n_lines + 1, window)
        if fmt == "f64":
            got = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines + 1, hop=hop, window=window, out_fmt=sa.OUT_DB20_F64)
            check_fp64(got[:-1], ref[:-1])
        else:
            got = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines + 1, hop=hop, window=window)
            check_fp32(got[:-1], ref[:-1], nfft)
        assert np.all(got[-1] == -150.0)


def test_compute_magnitudes_at_ui_maximum(svc, oracle):
    # main-scene.fxml:129-132: the NFFT slider tops out at 2^16
    iq = oracle.synth_iq("ci16_le", 2, 0, 65536)
    got = svc.compute_magnitudes(iq, 0, 65536, "ci16_le")
    check_fp64(got[None, :], oracle.compute_magnitudes(iq, 0, 65536, "ci16_le")[None, :])
    with pytest.raises(NotImplementedError):
        svc.compute_magnitudes(np.zeros(8 << 17, np.uint8), 0, 1 << 17, "cf32_le")


# ---- the whole power-of-two range of the reference service (commons-math3 takes any 2^k) ----------
@pytest.mark.parametrize("nfft", [2, 4, 8, 16, 32])
@pytest.mark.parametrize("datatype", ["cf32_le", "ci16_be", "cu8", "cf64_le"])
def test_tiny_nfft(svc, oracle, datatype, nfft):
    n_lines = 37
    iq = oracle.synth_iq(datatype, nfft, 0, n_lines * nfft)
    ref = oracle.waterfall(iq, 0, datatype, nfft, nfft, n_lines)
    check_fp32(svc.compute_waterfall(iq, 0, nfft, datatype, n_lines), ref, max(nfft, 4))
    one = svc.compute_magnitudes(iq, 0, nfft, datatype)
    check_fp64(one[None, :], oracle.compute_magnitudes(iq, 0, nfft, datatype, cf64_decode=True)[None, :])


# ---- Welch for everything the fused path does not take (cf64, big endian, very small / large nfft) --
@pytest.mark.parametrize("datatype,nfft,hop,n_seg", [("cf64_le", 1024, 512, 9), ("cf64_be", 16384, 4096, 6),
                                                     ("ci16_be", 2048, 1024, 7), ("cf32_le", 64, 16, 40),
                                                     ("cf32_le", 32768, 8192, 5), ("cf64_le", 65536, 32768, 3)])
def test_welch_fallback_paths(svc, oracle, datatype, nfft, hop, n_seg):
    fs = 1e6
    iq = oracle.synth_iq(datatype, 8, 0, (n_seg - 1) * hop + nfft)
    f_ref, p_ref = oracle.welch_psd(iq, 0, datatype, nfft, hop, n_seg, oracle.WIN_HANN, oracle.PSD_DENSITY, fs)
    f, p = svc.welch_psd(iq, 0, datatype, fs, nfft=nfft, hop=hop, n_seg=n_seg)
    assert np.array_equal(f, f_ref)
    tol = 1e-6 if datatype.startswith("cf64") else 5e-6        # the result itself is float32
    assert np.abs(p[0] - p_ref).max() <= tol * p_ref.max()


def test_calculate_psd_welch_call_shape(svc, oracle):
    # AnalysisDialogController.java:303-313: data = double[2][N], nfft = 8192 (or the length if shorter)
    import scipy.signal
    rng = np.random.default_rng(5)
    n, fs, nfft = 50000, 250e3, 8192
    x = rng.normal(size=n) + 1j * rng.normal(size=n) + 3 * np.exp(2j * np.pi * 0.11 * np.arange(n))
    data = np.stack([x.real, x.imag])
    out = svc.calculate_psd_welch(data, fs, nfft)
    assert out.shape == (2, nfft)
    f_ref, p_ref = scipy.signal.welch(x, fs, window="hann", nperseg=nfft, noverlap=nfft // 2, nfft=nfft,
                                      detrend=False, return_onesided=False, scaling="density")
    assert np.allclose(out[0], np.fft.fftshift(f_ref))
    # row 1 is what the caller reads -- decibels (ADC:319-328 additive dB offset, :612/:626 "%.1f dB" labels,
    # :675 SNR = difference, :751 "dB/Hz"): 10 log10(P + 1e-20), the floor of include/specgpu.h
    db_ref = 10 * np.log10(np.fft.fftshift(p_ref) + 1e-20)
    assert np.abs(out[1] - db_ref).max() <= 1e-5            # dB; 1e-6 relative in power is 4.3e-6 dB
    assert int(np.argmax(out[1])) == int(round(0.11 * nfft)) + nfft // 2
    lin = svc.calculate_psd_welch(data, fs, nfft, db=False)  # the knob: linear density
    assert np.abs(lin[1] - np.fft.fftshift(p_ref)).max() <= 1e-6 * p_ref.max()
    assert np.abs(10 * np.log10(lin[1] + 1e-20) - out[1]).max() <= 1e-9
    with pytest.raises(IndexError):                      # shorter than nfft: the caller picks nfft (ADC:303-307)
        svc.calculate_psd_welch(data[:, :100], fs, nfft)
    with pytest.raises(ValueError):
        svc.calculate_psd_welch(data, fs, 0)


@pytest.mark.parametrize("n", [1, 2, 100, 1000, 4097, 8191])
def test_welch_any_length(svc, oracle, n):
    """ADC:303-307: a burst shorter than 8192 samples is handed to calculatePsdWelch with
    nfft = data[0].length -- any integer.  Plain fp64 DFT on the device; checked against the oracle's
    O(N^2) DFT and against numpy.fft.fft, <= 1e-9 of the peak (doubles out)."""
    rng = np.random.default_rng(n)
    fs = 48e3
    x = rng.normal(size=n) + 1j * rng.normal(size=n) + 2 * np.exp(2j * np.pi * 0.2 * np.arange(n))
    data = np.stack([x.real, x.imag])
    window = sa.WIN_RECT if n == 1 else sa.WIN_HANN          # a one-point Hann window is 0
    out = svc.calculate_psd_welch(data, fs, n, window=window, db=False)
    out_db = svc.calculate_psd_welch(data, fs, n, window=window)          # the default: the dB row the dialog reads
    assert out.shape == (2, n) and out.dtype == np.float64
    assert np.array_equal(out_db[0], out[0])
    assert np.abs(out_db[1] - 10 * np.log10(out[1] + 1e-20)).max() <= 1e-9
    iq = np.ascontiguousarray(data.T).astype("<f8").tobytes()
    f_ref, p_ref = oracle.welch_psd(np.frombuffer(iq, np.uint8), 0, "cf64_le", n, max(n // 2, 1), 1, window,
                                    oracle.PSD_DENSITY, fs)
    w = np.ones(n) if window == sa.WIN_RECT else 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)
    p_np = np.fft.fftshift(np.abs(np.fft.fft(x * w)) ** 2 / (fs * np.sum(w * w)))
    assert np.array_equal(out[0], f_ref) and np.allclose(out[0], np.fft.fftshift(np.fft.fftfreq(n, 1 / fs)), atol=1e-9)
    assert np.abs(out[1] - p_ref).max() <= 1e-9 * p_ref.max()
    assert np.abs(out[1] - p_np).max() <= 1e-9 * p_np.max()
    if n == 1:
        with pytest.raises(ValueError):                      # Hann of one point sums to zero
            svc.calculate_psd_welch(data, fs, 1)


def test_welch_any_length_raw_bytes_and_segments(svc, oracle):
    # the batched entry over raw recording bytes takes the same lengths: several segments, two PSDs, floats out
    dt, nfft, hop, n_seg, fs = "ci16_le", 1500, 700, 4, 1e6
    per = (n_seg - 1) * hop + nfft
    iq = oracle.synth_iq(dt, 21, 0, 2 * per)
    f, p = svc.welch_psd(iq, 0, dt, fs, nfft=nfft, hop=hop, n_seg=n_seg, n_psd=2, psd_stride_bytes=per * 4)
    for b in range(2):
        f_ref, ref = oracle.welch_psd(iq, b * per * 4, dt, nfft, hop, n_seg, oracle.WIN_HANN, oracle.PSD_DENSITY, fs)
        assert np.array_equal(f, f_ref) and np.abs(p[b] - ref).max() <= 1e-6 * ref.max()
    with pytest.raises(NotImplementedError):
        svc.welch_psd(iq, 0, dt, fs, nfft=65537, hop=1, n_seg=1)


def test_context_shared_by_threads(svc, oracle):
    """AsyncExtractDownConvertService.java:27-35,52-55 runs the singleton on a pool of availableProcessors()
    threads: eight host threads on ONE context give the serial results."""
    import threading
    edc = sa.ExtractDownConvertService(svc)
    raw = oracle.synth_iq("ci16_le", 77, 0, 300000)
    jobs = [(1000 * i, 40000 + 1111 * i, 0.01 * (i + 1), 4 + i, bool(i & 1)) for i in range(8)]
    serial = [edc.extract_and_down_convert(raw, s0, cnt, "ci16_le", f, d, fast) for (s0, cnt, f, d, fast) in jobs]
    lines = [svc.compute_magnitudes(raw, 4 * 1024 * i, 1024, "ci16_le") for i in range(8)]
    got, got_lines, errors = [None] * 8, [None] * 8, []

    def work(i):
        try:
            for _ in range(5):
                s0, cnt, f, d, fast = jobs[i]
                got[i] = edc.extract_and_down_convert(raw, s0, cnt, "ci16_le", f, d, fast)
                got_lines[i] = svc.compute_magnitudes(raw, 4 * 1024 * i, 1024, "ci16_le")
        except Exception as e:                               # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for i in range(8):
        assert np.array_equal(got[i], serial[i]) and np.array_equal(got_lines[i], lines[i])


def test_huge_counts_are_rejected_before_any_launch(svc):
    """Range checks must not wrap: counts near 2^64 are SPEC_ERANGE / SPEC_EINVAL, never a launch."""
    import ctypes as C
    import torch
    from spectral_analyzer_amd import _lib as L
    lib, ctx = L.load(), svc._ctx
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    out = torch.zeros(4096, dtype=torch.float64, device="cuda")
    big = 2 ** 64 - 1
    p, o = buf.data_ptr(), out.data_ptr()
    assert lib.spec_extract_iq(ctx, p, 1, 4096, 1, big, L.DT_CI16_LE, o, o, 1) == L.SPEC_ERANGE
    assert lib.spec_extract_iq(ctx, p, 1, 4096, big, 2, L.DT_CI16_LE, o, o, 1) == L.SPEC_ERANGE
    assert lib.spec_extract_iq(ctx, p, 1, 4096, 2 ** 62, 2 ** 62, L.DT_CF64_LE, o, o, 1) == L.SPEC_ERANGE
    assert lib.spec_down_convert(ctx, p, 1, 4096, 1, big, L.DT_CI16_LE, 0.1, 2, 0, o, o, 1) == L.SPEC_ERANGE
    assert lib.spec_down_convert(ctx, p, 1, 4096, 1, big - 7, L.DT_CF32_LE, 0.1, 3, 1, o, o, 1) == L.SPEC_ERANGE
    assert lib.spec_magnitude_trace(ctx, o, o, 1, big // 8, 0.5, o, 1) == L.SPEC_ERANGE
    assert lib.spec_inst_freq_trace(ctx, o, o, 1, 2 ** 61, 0.5, 1.0, 0.0, o, 1) == L.SPEC_ERANGE
    f32 = torch.zeros(1024, dtype=torch.float32, device="cuda")
    assert lib.spec_welch_psd(ctx, p, 1, 4096, 0, big // 2, 2 ** 32 - 1, L.DT_CF32_LE, 64, 2 ** 32 - 1, 2 ** 32 - 1,
                              L.WIN_HANN, L.PSD_DENSITY, 1.0, 0, None, f32.data_ptr(), 1) == L.SPEC_ERANGE
    assert lib.spec_welch_psd(ctx, p, 1, 4096, 0, 16, 3, L.DT_CF32_LE, 64, 2 ** 32 - 1, 2 ** 32 - 1,
                              L.WIN_HANN, L.PSD_DENSITY, 1.0, 0, None, f32.data_ptr(), 1) == L.SPEC_ERANGE
    assert lib.spec_waterfall(ctx, p, 1, 4096, 0, L.DT_CF32_LE, 64, 64, big, L.WIN_RECT, L.OUT_DB20_F32, -150.0,
                              f32.data_ptr(), 1) == L.SPEC_ERANGE
    assert lib.spec_welch_psd_planar_f64(ctx, o, o, 1, 2 ** 63, 64, 1, L.WIN_HANN, L.PSD_DENSITY, 1.0, 0, None,
                                         o) in (L.SPEC_ERANGE, L.SPEC_EINVAL)
    torch.cuda.synchronize()
    assert float(out.abs().sum()) == 0.0 and float(f32.abs().sum()) == 0.0     # nothing was written


def test_calls_leave_the_current_device_alone(svc):
    import torch
    before = torch.cuda.current_device()
    svc.compute_magnitudes(np.zeros(8 * 64, np.uint8), 0, 64, "cf32_le")
    assert torch.cuda.current_device() == before


# ---- renderSpectrogram + getColorForMagnitude (MC:1261-1291, MC:926-957) ------
@pytest.mark.parametrize("colormap", [sa.CMAP_GRAYSCALE, sa.CMAP_HEATMAP])
def test_render_is_bit_exact_on_the_same_tile(svc, oracle, colormap):
    rng = np.random.default_rng(colormap)
    W, N, H, fs = 173, 2048, 611, 1e6
    tile = rng.uniform(-170, 60, size=(W, N)).astype(np.float32)
    tile[5] = -150.0                                     # an EOF line
    ref = oracle.render_spectrogram(tile.astype(np.float64), H, fs, -110.0, -15.0, colormap)
    got = svc.render_spectrogram(tile, H, fs, -110.0, -15.0, colormap)
    assert got.shape == (H, W, 4) and np.array_equal(got, ref)
    import torch
    got_d = svc.render_spectrogram(torch.from_numpy(tile).cuda(), H, fs, -110.0, -15.0, colormap)
    torch.cuda.synchronize()
    assert np.array_equal(got_d.cpu().numpy(), ref)


def test_waterfall_render_end_to_end(svc, oracle):
    # one redraw of MainController.updateDisplay(): canvasW lines at hop = nfft, then the image
    dt, nfft, W, H, fs = "ci16_le", 1024, 200, 257, 2e6
    iq = oracle.synth_iq(dt, 4, 0, (W - 3) * nfft)       # the last 3 columns run past the end -> -150 dB
    wf = oracle.waterfall(iq, 0, dt, nfft, nfft, W)
    for cmap in (sa.CMAP_GRAYSCALE, sa.CMAP_HEATMAP):
        ref = oracle.render_spectrogram(wf, H, fs, -100.0, 0.0, cmap).astype(np.int16)
        got = svc.waterfall_render(iq, 0, nfft, dt, W, H, fs, colormap=cmap).astype(np.int16)
        d = np.abs(got - ref)
        assert d.max() <= 1, "more than one 8-bit step"   # fp32 dB vs fp64 dB can flip a rounding
        assert (d > 0).mean() < 2e-3
    with pytest.raises(ValueError):
        svc.render_spectrogram(np.zeros((4, 64), np.float32), 10, 1e6, colormap=7)


def test_cpp_host_example(tmp_path):
    """The C++ mirror (include/specgpu.hpp) driven by a plain g++ program."""
    import subprocess
    from spectral_analyzer_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "example")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "integration", "cpp", "example.cpp"), "-L" + libdir, "-lspecgpu",
                           "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "example ok" in out.stdout, out.stdout + out.stderr


def test_two_contexts_two_streams(oracle):
    """Distinct contexts are independent (one per host thread / stream): run two from two threads."""
    import threading
    import torch
    results = {}

    def work(name, datatype, nfft):
        st = torch.cuda.Stream()
        with sa.SpectralService(0, stream=st.cuda_stream) as s:
            iq = oracle.synth_iq(datatype, 5, 0, 64 * nfft)
            d = torch.from_numpy(iq).cuda()
            outs = [s.compute_waterfall(d, 0, nfft, datatype, 127, hop=nfft // 2) for _ in range(20)]
            s.synchronize()
            results[name] = (outs[-1].cpu().numpy(), oracle.waterfall(iq, 0, datatype, nfft, nfft // 2, 127), nfft)

    th = [threading.Thread(target=work, args=("a", "cf32_le", 4096)),
          threading.Thread(target=work, args=("b", "ci16_le", 1024))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for got, ref, nfft in results.values():
        check_fp32(got, ref, nfft)


# ---- host-buffer pipeline (two-deep staging, copy-in / kernels / copy-out overlapped) -----------------
@pytest.mark.parametrize("datatype,nfft,hop", [("ci16_le", 1024, 512), ("cf32_le", 4096, 4096), ("cf64_le", 256, 100)])
@pytest.mark.parametrize("chunk_mb", [1, 64])
def test_host_pipeline_matches_device_path(svc, datatype, nfft, hop, chunk_mb):
    """Many small chunks (1 MiB) force every slot / event / helper-thread hand-off of the staged path;
    the result must equal the single-launch device-resident path bit for bit, EOF lines included."""
    import torch
    n_samples = 700_000
    iq = svc.synth_iq(datatype, 77, 0, n_samples)
    n_lines = (n_samples - nfft) // hop + 1 + 3   # + three lines past the end (MC:994-998)
    fmt = sa.OUT_DB20_F64 if datatype.startswith("cf64") else sa.OUT_DB20_F32
    ref = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines, hop=hop, out_fmt=fmt)
    torch.cuda.synchronize()
    svc.set_option("stage_chunk_mb", chunk_mb)
    try:
        host = iq.cpu().numpy()
        got = svc.compute_waterfall(host, 0, nfft, datatype, n_lines, hop=hop, out_fmt=fmt)      # host -> host
        assert np.array_equal(got, ref.cpu().numpy())
        dev_out = torch.empty_like(ref)
        s = svc
        s._check(s._lib.spec_waterfall(s._ctx, host.ctypes.data, 0, host.size, 0, sa.dtype_from_sigmf(datatype), nfft, hop,
                                       n_lines, sa.WIN_RECT, fmt, -150.0, dev_out.data_ptr(), 1))  # host -> device
        torch.cuda.synchronize()
        assert torch.equal(dev_out, ref)
        out_h = np.empty(ref.shape, dtype=got.dtype)
        s._check(s._lib.spec_waterfall(s._ctx, iq.data_ptr(), 1, iq.numel(), 0, sa.dtype_from_sigmf(datatype), nfft, hop,
                                       n_lines, sa.WIN_RECT, fmt, -150.0, out_h.ctypes.data, 0))   # device -> host
        assert np.array_equal(out_h, got)
    finally:
        svc.set_option("stage_chunk_mb", 64)


# ---- read-ahead of the per-slice call (the unmodified MainController loop, MC:982-993) --------------
@pytest.mark.parametrize("datatype,nfft", [("ci16_le", 1024), ("cf32_le", 4096), ("cu8", 256), ("cf64_be", 512)])
def test_compute_magnitudes_readahead_is_transparent(svc, oracle, datatype, nfft):
    """Walk a buffer slice by slice as the reference loop does: the batched read-ahead must return exactly
    what one launch per call returns, notice bytes that changed under it, and keep the error behaviour."""
    bps = oracle.bytes_per_sample(datatype)
    n_slices = 700
    iq = oracle.synth_iq(datatype, 5, 0, n_slices * nfft + 17).copy()
    svc.set_option("readahead_lines", 0)
    plain = [svc.compute_magnitudes(iq, t * nfft * bps, nfft, datatype) for t in range(0, n_slices, 37)]
    svc.set_option("readahead_lines", 256)
    try:
        walked = [svc.compute_magnitudes(iq, t * nfft * bps, nfft, datatype) for t in range(n_slices)]
        for k, t in enumerate(range(0, n_slices, 37)):
            assert np.array_equal(walked[t], plain[k]), t
        # bytes change between two calls of a walk: the cached line must not be served
        a = svc.compute_magnitudes(iq, 10 * nfft * bps, nfft, datatype)
        b = svc.compute_magnitudes(iq, 11 * nfft * bps, nfft, datatype)      # sequential: batch computed here
        iq[12 * nfft * bps: 13 * nfft * bps] = oracle.synth_iq(datatype, 99, 0, nfft)
        c = svc.compute_magnitudes(iq, 12 * nfft * bps, nfft, datatype)
        assert np.array_equal(a, walked[10]) and np.array_equal(b, walked[11]) and not np.array_equal(c, walked[12])
        check_fp64(c[None, :], oracle.compute_magnitudes(iq, 12 * nfft * bps, nfft, datatype, cf64_decode=True)[None, :])
        d = svc.compute_magnitudes(iq, 13 * nfft * bps, nfft, datatype)      # unchanged slice, still cached
        assert np.array_equal(d, walked[13])
        # the walk runs into the end of the buffer: last whole slice fine, the next one out of range
        last = (iq.size // (nfft * bps)) - 1
        for t in range(last - 3, last + 1):
            svc.compute_magnitudes(iq, t * nfft * bps, nfft, datatype)
        with pytest.raises(IndexError):
            svc.compute_magnitudes(iq, (last + 1) * nfft * bps, nfft, datatype)
    finally:
        svc.set_option("readahead_lines", 256)


# ---- fused render: only the bins the image samples leave the FFT kernel -------------------------------
@pytest.mark.parametrize("datatype,nfft,hop,height", [("cf32_le", 4096, 4096, 600), ("cf32_le", 4096, 2048, 4096),
                                                       ("ci16_le", 1024, 1024, 333), ("cu8", 256, 256, 256),
                                                       ("ci16_be", 2048, 1024, 1000), ("ci8", 512, 300, 17),
                                                       ("cu8", 256, 256, 300),        # more rows than bins: two-pass form
                                                       ("cf32_le", 8192, 8192, 700),  # 32-point threads: two-pass form
                                                       ("cf64_le", 1024, 512, 400)])  # fp64 arithmetic: two-pass form
@pytest.mark.parametrize("colormap", [sa.CMAP_GRAYSCALE, sa.CMAP_HEATMAP])
def test_waterfall_render_fused_equals_two_pass(svc, oracle, datatype, nfft, hop, height, colormap):
    """spec_waterfall_render with the compact tile (default) must give the very pixels of the two-pass form
    (full dB tile, then the colour kernel), EOF columns included, and both match the restated Java renderer
    applied to the GPU's own dB tile."""
    width = 150
    n = (width - 4) * hop + nfft                      # the last three columns run past the end: -150 dB
    iq = oracle.synth_iq(datatype, 13, 0, n)
    fs, lo, hi = 2.0e6, -120.0, -20.0
    fused = svc.waterfall_render(iq, 0, nfft, datatype, width, height, fs, min_db=lo, max_db=hi, colormap=colormap, hop=hop)
    svc.set_option("render_fused", 0)
    try:
        plain = svc.waterfall_render(iq, 0, nfft, datatype, width, height, fs, min_db=lo, max_db=hi, colormap=colormap, hop=hop)
    finally:
        svc.set_option("render_fused", 1)
    assert fused.shape == (height, width, 4) and np.array_equal(fused, plain)
    tile = svc.compute_waterfall(iq, 0, nfft, datatype, width, hop=hop)
    assert np.array_equal(fused, oracle.render_spectrogram(tile.astype(np.float64), height, fs, lo, hi, colormap))


def test_waterfall_render_fused_device_resident(svc, oracle):
    import torch
    nfft, width, height = 4096, 512, 700
    iq = svc.synth_iq("cf32_le", 21, 0, width * nfft)
    img = svc.waterfall_render(iq, 0, nfft, "cf32_le", width, height, 1e6)
    svc.set_option("render_fused", 0)
    try:
        ref = svc.waterfall_render(iq, 0, nfft, "cf32_le", width, height, 1e6)
    finally:
        svc.set_option("render_fused", 1)
    torch.cuda.synchronize()
    assert img.is_cuda and torch.equal(img, ref)


# ---- spec_waterfall_multi: one waterfall over several contexts (SURVEY 8e at the C ABI) ------------------------
@pytest.mark.parametrize("datatype,nfft,hop,n_lines,fmt", [("cf32_le", 4096, 2048, 1001, sa.OUT_DB20_F32),
                                                           ("ci16_le", 1024, 1024, 77, sa.OUT_DB20_F32),
                                                           ("cf64_le", 16384, 8192, 60, sa.OUT_DB20_F64),   # the single-workgroup fp64 kernel
                                                           ("cf64_le", 32768, 16384, 40, sa.OUT_DB20_F64),  # < 64 lines: the two-launch four-step path in every context (three persistent team kernels cannot share ONE device)
                                                           ("cu8", 256, 100, 2, sa.OUT_POW_F32)])
def test_waterfall_multi_equals_the_single_context_tile(svc, oracle, datatype, nfft, hop, n_lines, fmt):
    """Three contexts (all on device 0 on the one-GPU box; one per GPU on a node), one host thread each inside the
    library, contiguous line ranges with the nfft - hop halo exactly as dist.shard_lines / shard_span: the tile is the
    single-context tile BIT FOR BIT -- host buffer -> host tile (what the JVM host uses), and device-resident
    shards -> a tile on the consumer's device, the peers' pieces sent with hipMemcpyPeerAsync behind their kernels.
    Lines past the end are -150 (MC:994-998); a shard may be empty (fewer lines than contexts)."""
    import torch
    from spectral_analyzer_amd import dist as sd
    bps = oracle.bytes_per_sample(datatype)
    iq = oracle.synth_iq(datatype, 5, 0, (n_lines - 1) * hop + nfft)
    extra = 3                                                               # lines past the end of the recording
    one = svc.compute_waterfall(iq, 0, nfft, datatype, n_lines + extra, hop=hop, out_fmt=fmt)
    peers = [sa.SpectralService(0, stream=torch.cuda.Stream().cuda_stream) for _ in range(2)]
    services = [svc] + peers
    try:
        host = sa.compute_waterfall_multi(services, iq, 0, nfft, datatype, n_lines + extra, hop=hop, out_fmt=fmt)
        assert host.dtype == one.dtype and np.array_equal(host, one)
        assert np.all(host[n_lines:] == -150.0)
        # device-resident shards -> device tile on services[0]'s device, in 5 pieces per peer
        shards = []
        for r in range(3):
            l0, l1 = sa.shard_lines(n_lines, 3, r)
            assert (l0, l1) == sd.shard_lines(n_lines, 3, r)
            first, nb = sa.shard_span(l0, l1, datatype, nfft, hop)
            assert (first, nb) == tuple(bps * x for x in sd.shard_span(l0, l1, nfft, hop))
            shards.append(torch.from_numpy(iq[first:first + nb].copy()).cuda() if nb else None)
        out = torch.full((n_lines + extra, nfft), float("nan"), dtype=torch.from_numpy(one[:1]).dtype, device="cuda")
        got = sa.compute_waterfall_multi(services, shards, 0, nfft, datatype, n_lines + extra, hop=hop, out_fmt=fmt, out=out,
                                         n_bytes=iq.size, n_chunks=5)
        assert got is out and np.array_equal(out.cpu().numpy(), one)
        # host buffer -> device tile (the peers stage their spans, then send)
        out.fill_(float("nan"))
        sa.compute_waterfall_multi(services, iq, 0, nfft, datatype, n_lines + extra, hop=hop, out_fmt=fmt, out=out)
        assert np.array_equal(out.cpu().numpy(), one)
        # "multi_verify": every piece checksummed on the peer's device before it leaves and where it landed on the
        # consumer's; set on ctx[0] it covers every peer.  The peers report the path their copies took.
        svc.set_option("multi_verify", 1)
        try:
            out.fill_(float("nan"))
            sa.compute_waterfall_multi(services, shards, 0, nfft, datatype, n_lines + extra, hop=hop, out_fmt=fmt, out=out,
                                       n_bytes=iq.size, n_chunks=5)
            assert np.array_equal(out.cpu().numpy(), one)
            for r, p in enumerate(peers, start=1):
                l0, l1 = sa.shard_lines(n_lines, 3, r)
                assert p.get_option("multi_verified") == min(5, l1 - l0)    # pieces of this peer, all agreed
                assert p.get_option("multi_peer_access") == 2               # consumer on the same device (one-GPU box)
            assert svc.get_option("multi_verified") == 0 and svc.get_option("multi_peer_access") == -1   # the consumer sends nothing
            # and it does notice: one word of peer 2's last piece damaged where it landed ("multi_verify_corrupt", tests only)
            if sa.shard_lines(n_lines, 3, 2)[1] > sa.shard_lines(n_lines, 3, 2)[0]:
                peers[1].set_option("multi_verify_corrupt", 1)
                with pytest.raises(RuntimeError, match="multi_verify: piece .* does not match what landed .* same device"):
                    sa.compute_waterfall_multi(services, shards, 0, nfft, datatype, n_lines + extra, hop=hop, out_fmt=fmt, out=out,
                                               n_bytes=iq.size, n_chunks=5)
        finally:
            svc.set_option("multi_verify", 0)
        with pytest.raises(ValueError):                                     # the same context twice
            sa.compute_waterfall_multi([svc, svc], iq, 0, nfft, datatype, n_lines, hop=hop, out_fmt=fmt)
        with pytest.raises(ValueError):                                     # the reference's error behaviour is kept
            sa.compute_waterfall_multi(services, iq, 0, nfft - 1, datatype, n_lines, hop=hop, out_fmt=fmt)
    finally:
        for p in peers:
            p.close()


def test_last_error_is_not_inherited_from_a_destroyed_context(oracle):
    """spec_last_error's per-thread cache is keyed on a per-context generation id: a context created after another was
    destroyed -- quite possibly at the same address -- starts with an empty error text."""
    import torch
    iq = oracle.synth_iq("cf32_le", 1, 0, 4096)
    for _ in range(8):
        a = sa.SpectralService(0, stream=torch.cuda.Stream().cuda_stream)
        with pytest.raises(ValueError, match="power of two"):
            a.compute_waterfall(iq, 0, 1000, "cf32_le", 1)
        addr = a._ctx.value
        a.close()
        b = sa.SpectralService(0, stream=torch.cuda.Stream().cuda_stream)
        try:
            assert b._lib.spec_last_error(b._ctx) == b"", (hex(addr), hex(b._ctx.value))
        finally:
            b.close()


def test_last_error_of_one_context_does_not_disturb_another(oracle):
    """Round-3 advisor: asking about context B used to overwrite this thread's recorded failure on context A (one
    thread-local cache) and to invalidate the text an earlier spec_last_error(A) had returned."""
    import threading
    import torch
    iq = oracle.synth_iq("cf32_le", 1, 0, 4096)
    a = sa.SpectralService(0, stream=torch.cuda.Stream().cuda_stream)
    b = sa.SpectralService(0, stream=torch.cuda.Stream().cuda_stream)
    try:
        with pytest.raises(ValueError, match="power of two"):
            a.compute_waterfall(iq, 0, 1000, "cf32_le", 1)                  # this thread's failure on A
        def other():                                                        # another thread fails on B
            with pytest.raises(ValueError, match="hop"):
                b.compute_waterfall(iq, 0, 1024, "cf32_le", 1, hop=0)
        t = threading.Thread(target=other); t.start(); t.join()
        text_a = a._lib.spec_last_error(a._ctx)
        assert b"power of two" in text_a
        assert b"hop" in b._lib.spec_last_error(b._ctx)                     # a COPY of the other thread's text ...
        assert a._lib.spec_last_error(a._ctx) == text_a and b"power of two" in a._lib.spec_last_error(a._ctx)   # ... and A's is intact
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("datatype,nfft,hop,n_seg,n_psd", [("cf32_le", 1024, 512, 6, 7), ("ci16_le", 4096, 1024, 9, 600),
                                                           ("cf64_le", 2048, 2048, 3, 5), ("cu8", 1000, 300, 4, 2),
                                                           ("cf32_le", 256, 64, 33, 1)])
def test_welch_multi_equals_the_single_context_batch(svc, oracle, datatype, nfft, hop, n_seg, n_psd):
    """spec_welch_psd_multi: the PSDs of a batch sharded over three contexts (all on device 0 on the one-GPU box) -- host
    buffer -> host result, and device-resident shards -> a result on the consumer's device (the peers' rows sent with
    hipMemcpyPeerAsync) -- are what one context returns for the batch: bit for bit while both take the same form of the kernel,
    to the rounding of the fp32 segment sums otherwise (a batch of >= two PSDs per CU is summed in one pass by the workgroup
    that walks the PSD, smaller ones in runs of <= 64 segments: 600 PSDs in one context against 200 in each of three); a
    shard may be empty (fewer PSDs than contexts); linear and dB."""
    import torch
    bps = oracle.bytes_per_sample(datatype)
    per = (n_seg - 1) * hop + nfft + 5
    start = 3
    if n_psd > 50:
        iq = svc.synth_iq(datatype, 31, 0, start + per * n_psd).cpu().numpy()
    else:
        iq = oracle.synth_iq(datatype, 31, 0, start + per * n_psd)
    fs = 3.0e6
    peers = [sa.SpectralService(0, stream=torch.cuda.Stream().cuda_stream) for _ in range(2)]
    services = [svc] + peers

    def same(got, one, db):
        if n_psd <= 50:
            return np.array_equal(got, one)
        if db:
            return np.abs(got - one).max() <= 1e-4
        return np.abs(got - one).max() <= 2e-6 * one.max()

    try:
        for db in (False, True):
            f1, one = svc.welch_psd(iq, start * bps, datatype, fs, nfft=nfft, hop=hop, n_seg=n_seg, n_psd=n_psd, psd_stride_bytes=per * bps, db=db)
            f, host = sa.welch_psd_multi(services, iq, start * bps, datatype, fs, nfft, hop, n_seg, n_psd, per * bps, db=db)
            assert np.array_equal(f, f1) and host.dtype == one.dtype and same(host, one, db)
            shards = []
            for r in range(3):
                a, b = sa.shard_lines(n_psd, 3, r)
                lo, hi = (start + a * per) * bps, (start + (b - 1) * per + (n_seg - 1) * hop + nfft) * bps
                shards.append(torch.from_numpy(iq[lo:hi].copy()).cuda() if b > a else None)
            out = torch.full((n_psd, nfft), float("nan"), dtype=torch.float32, device="cuda")
            _, got = sa.welch_psd_multi(services, shards, 0, datatype, fs, nfft, hop, n_seg, n_psd, per * bps, db=db, out=out)
            assert got is out and same(out.cpu().numpy(), one, db)
            peers[0].set_option("multi_verify", 1)                          # on ONE peer: its rows only
            try:
                out.fill_(float("nan"))
                sa.welch_psd_multi(services, shards, 0, datatype, fs, nfft, hop, n_seg, n_psd, per * bps, db=db, out=out)
                assert same(out.cpu().numpy(), one, db)
                a, b = sa.shard_lines(n_psd, 3, 1)
                assert peers[0].get_option("multi_verified") == (1 if b > a else 0) and peers[1].get_option("multi_verified") == 0
            finally:
                peers[0].set_option("multi_verify", 0)
        with pytest.raises(ValueError):                                     # the same context twice
            sa.welch_psd_multi([svc, svc], iq, 0, datatype, fs, nfft, hop, n_seg, n_psd, per * bps)
        with pytest.raises(IndexError):                                     # spec_welch_psd's range error, from a shard
            sa.welch_psd_multi(services, iq, 0, datatype, fs, nfft, hop, n_seg * 1000, n_psd, per * bps)
    finally:
        for p in peers:
            p.close()
