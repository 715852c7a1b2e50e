"""Python face of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Two independent restatements of the reference hot path live here:

* ``C``   -- ctypes bindings to ``oracle/libspec_oracle.so`` (spec_oracle.c), the
  fp64 restatement of ``SpectralService.computeMagnitudes``
  (services/SpectralService.java:33-85) -- with commons-math3 3.6.1's transform restated as
  published (bit-reversal shuffle, 4-term first stage, twiddles by recurrence; ``FFT_CM3``) and
  an exact-twiddle transform beside it as the accuracy yardstick (``FFT_EXACT``) -- and of the
  ``MainController.updateDisplay`` line loop (controllers/MainController.java:980-999).
* ``np_*`` -- a numpy restatement that uses ``numpy.fft.fft`` (a different FFT
  implementation) so the two can be checked against each other.

Parity status: "parity unpinned" against an execution of the Java reference (no
JVM / jars available); pinned against analytic known-answer tests and the
numpy / scipy cross-checks in tests/test_oracle.py.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libspec_oracle.so")

WIN_RECT, WIN_HANN = 0, 1
FFT_CM3, FFT_EXACT = 0, 1   # the reference's commons-math3 transform / the exact-twiddle yardstick
PSD_DENSITY, PSD_SPECTRUM = 0, 1

_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "spec_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p, dp = C.c_void_p, C.c_void_p
        L.so_bytes_per_sample.argtypes = [C.c_char_p]
        L.so_is_big_endian.argtypes = [C.c_char_p]
        L.so_fft_forward.argtypes = [dp, dp, C.c_uint32]
        L.so_fft_forward_cm3.argtypes = [dp, dp, C.c_uint32]
        L.so_fft_forward_exact.argtypes = [dp, dp, C.c_uint32]
        L.so_complex_abs.argtypes = [C.c_double, C.c_double]
        L.so_complex_abs.restype = C.c_double
        L.so_waterfall_fft.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.c_uint32,
                                       C.c_uint32, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_int, dp]
        L.so_compute_magnitudes.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_char_p, C.c_int, dp]
        L.so_waterfall.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.c_uint32,
                                   C.c_uint32, C.c_uint64, C.c_int, C.c_double, C.c_int, dp]
        L.so_count_lines.argtypes = [C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint32, C.c_uint32]
        L.so_count_lines.restype = C.c_uint64
        L.so_welch_psd.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.c_uint32,
                                   C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_double, C.c_int,
                                   dp, dp]
        L.so_display_conversion.argtypes = [C.c_double, C.c_uint32]
        L.so_display_conversion.restype = C.c_double
        L.so_render_spectrogram.argtypes = [dp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                            C.c_double, C.c_int, u8p]
        L.so_synth_iq.argtypes = [u8p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64]
        L.so_time_waterfall.argtypes = [u8p, C.c_uint64, C.c_char_p, C.c_uint32, C.c_uint32,
                                        C.c_uint64, C.c_int, C.c_int, dp]
        L.so_time_waterfall.restype = C.c_double
        L.so_extract_iq.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, dp, dp]
        L.so_down_convert_taps.argtypes = [C.c_uint32, C.c_int, dp, C.c_void_p]
        L.so_down_convert_taps.restype = C.c_uint32
        L.so_down_convert.argtypes = [dp, dp, C.c_uint64, C.c_double, C.c_uint32, C.c_int, dp, dp]
        L.so_magnitude_trace.argtypes = [dp, dp, C.c_uint64, C.c_double, dp]
        L.so_inst_freq_trace.argtypes = [dp, dp, C.c_uint64, C.c_double, C.c_double, C.c_double, dp]
        _lib = L
    return _lib


def _bytes_view(buf) -> np.ndarray:
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf.view(np.uint8).reshape(-1)
    return np.ascontiguousarray(a)


def bytes_per_sample(datatype: str) -> int:
    return lib().so_bytes_per_sample(datatype.encode())


def is_big_endian(datatype: str) -> bool:
    return bool(lib().so_is_big_endian(datatype.encode()))


def fft_forward(x: np.ndarray, fft: int = FFT_CM3) -> np.ndarray:
    """FFT_CM3: commons-math3 3.6.1 as published (the reference's transform); FFT_EXACT: the yardstick."""
    re = np.ascontiguousarray(x.real, dtype=np.float64).copy()
    im = np.ascontiguousarray(x.imag, dtype=np.float64).copy()
    fn = lib().so_fft_forward_cm3 if fft == FFT_CM3 else lib().so_fft_forward_exact
    rc = fn(re.ctypes.data, im.ctypes.data, len(re))
    if rc:
        raise ValueError("nfft must be a power of two")
    return re + 1j * im


def complex_abs(re: float, im: float) -> float:
    """Complex.abs() of commons-math3 (SS:80)."""
    return float(lib().so_complex_abs(float(re), float(im)))


def compute_magnitudes(buf, start_byte: int, nfft: int, datatype: str, cf64_decode: bool = False) -> np.ndarray:
    """SpectralService.computeMagnitudes (SS:33-85); returns double[nfft]."""
    b = _bytes_view(buf)
    need = start_byte + nfft * bytes_per_sample(datatype)
    if start_byte < 0 or need > b.size:
        raise IndexError("IndexOutOfBoundsException: %d > %d" % (need, b.size))
    out = np.empty(nfft, dtype=np.float64)
    rc = lib().so_compute_magnitudes(b.ctypes.data, start_byte, nfft, datatype.encode(),
                                     int(cf64_decode), out.ctypes.data)
    if rc:
        raise ValueError("nfft must be a power of two")
    return out


def count_lines(capacity: int, start_byte: int, datatype: str, nfft: int, hop: int) -> int:
    return int(lib().so_count_lines(capacity, start_byte, datatype.encode(), nfft, hop))


def waterfall(buf, start_byte: int, datatype: str, nfft: int, hop: int, n_lines: int,
              window: int = WIN_RECT, eof_fill: float = -150.0, power: bool = False,
              cf64_decode: bool = True, fft: int = FFT_CM3) -> np.ndarray:
    """MC:980-999 around SS:33-85.  ``fft=FFT_CM3`` (default): the reference's own transform and
    ``Complex.abs()``; ``FFT_EXACT``: the exact-twiddle yardstick with ``hypot``."""
    b = _bytes_view(buf)
    out = np.empty((n_lines, nfft), dtype=np.float64)
    rc = lib().so_waterfall_fft(b.ctypes.data, b.size, start_byte, datatype.encode(), int(cf64_decode),
                                nfft, hop, n_lines, window, eof_fill, int(power), int(fft), out.ctypes.data)
    if rc:
        raise ValueError("bad nfft/hop")
    return out


def welch_psd(buf, start_byte: int, datatype: str, nfft: int, hop: int, n_seg: int,
              window: int = WIN_HANN, scaling: int = PSD_DENSITY, fs: float = 1.0,
              db: bool = False, cf64_decode: bool = True):
    b = _bytes_view(buf)
    f = np.empty(nfft, dtype=np.float64)
    p = np.empty(nfft, dtype=np.float64)
    rc = lib().so_welch_psd(b.ctypes.data, b.size, start_byte, datatype.encode(), int(cf64_decode),
                            nfft, hop, n_seg, window, scaling, fs, int(db), f.ctypes.data, p.ctypes.data)
    if rc:
        raise ValueError("bad welch arguments")
    return f, p


def display_conversion(fs: float, nfft: int) -> float:
    return float(lib().so_display_conversion(fs, nfft))


def render_spectrogram(waterfall: np.ndarray, height: int, fs: float, min_db: float = -100.0,
                       max_db: float = 0.0, colormap: int = 0) -> np.ndarray:
    """MC:1261-1291 + MC:926-957; waterfall [width, nfft] (dB) -> uint8 [height, width, 4] BGRA."""
    w = np.ascontiguousarray(waterfall, dtype=np.float64)
    out = np.empty((height, w.shape[0], 4), dtype=np.uint8)
    lib().so_render_spectrogram(w.ctypes.data, w.shape[0], w.shape[1], height, fs, min_db, max_db, colormap,
                                out.ctypes.data)
    return out


def synth_iq(datatype: str, seed: int, first_sample: int, n_samples: int) -> np.ndarray:
    """SURVEY 8(d) counter-based synthetic IQ, as raw file bytes (uint8)."""
    out = np.empty(n_samples * bytes_per_sample(datatype), dtype=np.uint8)
    rc = lib().so_synth_iq(out.ctypes.data, datatype.encode(), seed, first_sample, n_samples)
    if rc:
        raise ValueError("unsupported datatype " + datatype)
    return out


def time_waterfall(buf, datatype: str, nfft: int, hop: int, n_lines: int, window: int, threads: int):
    b = _bytes_view(buf)
    cs = C.c_double(0)
    t = lib().so_time_waterfall(b.ctypes.data, b.size, datatype.encode(), nfft, hop, n_lines,
                                window, threads, C.addressof(cs))
    return float(t), float(cs.value)


def extract_iq(buf, start_sample: int, count: int, datatype: str, ref_cf64_stride8: bool = False):
    """EDC:60-97 reader -> (re, im) float64 arrays."""
    b = _bytes_view(buf)
    re, im = np.empty(count, dtype=np.float64), np.empty(count, dtype=np.float64)
    rc = lib().so_extract_iq(b.ctypes.data, b.size, start_sample, count, datatype.encode(), int(ref_cf64_stride8),
                             re.ctypes.data, im.ctypes.data)
    if rc:
        raise IndexError("IndexOutOfBoundsException")
    return re, im


def down_convert_taps(down: int, mode: int):
    c = C.c_uint32(0)
    k = lib().so_down_convert_taps(down, mode, None, C.addressof(c))
    h = np.empty(k, dtype=np.float64)
    lib().so_down_convert_taps(down, mode, h.ctypes.data, C.addressof(c))
    return h, int(c.value)


def down_convert(re, im, freq_off: float, down: int, mode: int = 0):
    re = np.ascontiguousarray(re, dtype=np.float64)
    im = np.ascontiguousarray(im, dtype=np.float64)
    n_out = len(re) // down if down else 0
    ore, oim = np.empty(n_out, dtype=np.float64), np.empty(n_out, dtype=np.float64)
    rc = lib().so_down_convert(re.ctypes.data, im.ctypes.data, len(re), freq_off, down, mode,
                               ore.ctypes.data, oim.ctypes.data)
    if rc:
        raise ValueError("bad down-converter arguments")
    return ore, oim


def magnitude_trace(re, im, alpha: float) -> np.ndarray:
    re = np.ascontiguousarray(re, dtype=np.float64)
    im = np.ascontiguousarray(im, dtype=np.float64)
    out = np.empty(len(re), dtype=np.float64)
    lib().so_magnitude_trace(re.ctypes.data, im.ctypes.data, len(re), alpha, out.ctypes.data)
    return out


def inst_freq_trace(re, im, alpha: float, fs: float, center_freq: float = 0.0) -> np.ndarray:
    re = np.ascontiguousarray(re, dtype=np.float64)
    im = np.ascontiguousarray(im, dtype=np.float64)
    out = np.empty(max(len(re) - 1, 0), dtype=np.float64)
    lib().so_inst_freq_trace(re.ctypes.data, im.ctypes.data, len(re), alpha, fs, center_freq, out.ctypes.data)
    return out


# --------------------------------------------------------------------------
# numpy restatement (independent FFT implementation)
# --------------------------------------------------------------------------
def np_decode(buf, start_byte: int, n: int, datatype: str, cf64_decode: bool = True) -> np.ndarray:
    """SS:40-65 decode table (+cf64 per EDC:79-81) -> complex128[n]."""
    b = _bytes_view(buf)
    bo = ">" if is_big_endian(datatype) else "<"
    if datatype.startswith("ci16"):
        v = np.frombuffer(b, dtype=bo + "i2", count=2 * n, offset=start_byte).astype(np.float64) / 32768.0
    elif datatype.startswith("cf32"):
        v = np.frombuffer(b, dtype=bo + "f4", count=2 * n, offset=start_byte).astype(np.float64)
    elif datatype.startswith("cu8"):
        v = (np.frombuffer(b, dtype=np.uint8, count=2 * n, offset=start_byte).astype(np.float64) - 127.5) / 128
    elif datatype.startswith("ci8"):
        v = np.frombuffer(b, dtype=np.int8, count=2 * n, offset=start_byte).astype(np.float64) / 128
    elif datatype.startswith("cf64") and cf64_decode:
        v = np.frombuffer(b, dtype=bo + "f8", count=2 * n, offset=start_byte).astype(np.float64)
    else:
        v = np.zeros(2 * n)
    return v[0::2] + 1j * v[1::2]


def np_window(nfft: int, window: int) -> np.ndarray:
    if window == WIN_HANN:
        return 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(nfft) / nfft)
    return np.ones(nfft)


def np_spectrum(buf, start_byte: int, datatype: str, nfft: int, hop: int, n_lines: int,
                window: int = WIN_RECT) -> np.ndarray:
    """Unshifted complex spectra [n_lines, nfft] (all lines must be in range)."""
    bps = bytes_per_sample(datatype)
    w = np_window(nfft, window)
    rows = [np_decode(buf, start_byte + t * hop * bps, nfft, datatype) * w for t in range(n_lines)]
    return np.fft.fft(np.asarray(rows), axis=1)


def np_waterfall(buf, start_byte: int, datatype: str, nfft: int, hop: int, n_lines: int,
                 window: int = WIN_RECT, eof_fill: float = -150.0) -> np.ndarray:
    b = _bytes_view(buf)
    bps = bytes_per_sample(datatype)
    out = np.full((n_lines, nfft), eof_fill, dtype=np.float64)
    w = np_window(nfft, window)
    for t in range(n_lines):
        off = start_byte + t * hop * bps
        if off + nfft * bps <= b.size:
            X = np.fft.fft(np_decode(b, off, nfft, datatype) * w)
            out[t] = np.fft.fftshift(20 * np.log10(np.abs(X) + 1e-10))
    return out
