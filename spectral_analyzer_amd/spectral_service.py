"""Host-side mirror of the reference's ``SpectralService`` for the GPU path.

Reference (paths under src/main/java/net/kcundercover/spectral_analyzer/):
  * ``services/SpectralService.java:33-85``  computeMagnitudes -- one line
  * ``controllers/MainController.java:980-999``  the slice loop that calls it
  * ``controllers/AnalysisDialogController.java:303-313``  the Welch PSD call

The class keeps the reference's method name, argument order and error
behaviour (``compute_magnitudes(buffer, start_byte, nfft, datatype)``) and adds
the batched calls the device boundary wants (``compute_waterfall``,
``welch_psd``).  Everything numeric happens in libspecgpu.so through the C ABI
of ``include/specgpu.h``; PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib as L

_ERRORS = {
    L.SPEC_EINVAL: ValueError,            # IllegalArgumentException
    L.SPEC_ERANGE: IndexError,            # IndexOutOfBoundsException
    L.SPEC_EDEVICE: RuntimeError,
    L.SPEC_ENOMEM: MemoryError,
    L.SPEC_EUNSUPPORTED: NotImplementedError,
}

_OUT_NP = {L.OUT_DB20_F32: np.float32, L.OUT_POW_F32: np.float32,
           L.OUT_DB20_F64: np.float64, L.OUT_POW_F64: np.float64}


def dtype_from_sigmf(datatype: str) -> int:
    """SigMF datatype string -> spec_dtype (startsWith rules of SS:35-38)."""
    return int(L.load().spec_dtype_from_sigmf(datatype.encode()))


def bytes_per_sample(datatype: str) -> int:
    """``Global.getBytesPerSample()`` (sigmf/Global.java:67-79)."""
    return int(L.load().spec_bytes_per_sample(dtype_from_sigmf(datatype)))


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _host_bytes(buffer) -> np.ndarray:
    if isinstance(buffer, np.ndarray):
        a = buffer.reshape(-1).view(np.uint8)
    else:
        a = np.frombuffer(buffer, dtype=np.uint8)
    return np.ascontiguousarray(a)


def shard_lines(n_lines: int, n_shards: int, shard: int) -> Tuple[int, int]:
    """``spec_shard_lines``: the half-open line range of a shard (the partition of ``compute_waterfall_multi``)."""
    a, b = C.c_uint64(0), C.c_uint64(0)
    L.load().spec_shard_lines(int(n_lines), int(n_shards), int(shard), C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def shard_span(first_line: int, end_line: int, datatype: str, nfft: int, hop: int) -> Tuple[int, int]:
    """``spec_shard_span``: (first byte counted from the recording's start byte, byte count) of those lines."""
    a, b = C.c_uint64(0), C.c_uint64(0)
    L.load().spec_shard_span(int(first_line), int(end_line), dtype_from_sigmf(datatype), int(nfft), int(hop),
                             C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def compute_waterfall_multi(services, buffer, start_byte: int, nfft: int, datatype: str, n_lines: int,
                            hop: Optional[int] = None, window: int = L.WIN_RECT, out_fmt: int = L.OUT_DB20_F32,
                            eof_fill: float = -150.0, out=None, n_bytes: Optional[int] = None, n_chunks: int = 0):
    """``spec_waterfall_multi``: one waterfall (MC:980-999) with its lines sharded over several ``SpectralService``
    contexts -- one host thread per context inside the library, no ``torch.distributed``; what a single-process
    host (the reference is one JVM) uses to drive the GPUs of a node.

    ``buffer``: host bytes of the whole recording (every context stages its own span), or a LIST of CUDA uint8
    tensors, entry r on ``services[r]``'s device holding exactly shard r's span (``shard_lines`` / ``shard_span``
    of the first min(n_lines, whole lines in the recording) lines; ``n_bytes`` = the recording's byte count then).
    ``out``: None / numpy array -> host tile; a CUDA tensor on ``services[0]``'s device -> the peers send their
    pieces there with ``hipMemcpyPeerAsync`` behind each piece's kernels.  Returns the tile."""
    lib = L.load()
    hop = int(nfft if hop is None else hop)
    dt = dtype_from_sigmf(datatype)
    np_dt = _OUT_NP[out_fmt]
    n = len(services)
    ctxs = (C.c_void_p * n)(*[s._ctx for s in services])
    keep = None
    if isinstance(buffer, (list, tuple)):
        import torch
        if len(buffer) != n or n_bytes is None:
            raise ValueError("device input: one tensor per service and n_bytes (the recording's byte count)")
        for t in buffer:
            if t is not None and (not t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous()):
                raise ValueError("device buffers must be contiguous CUDA uint8 tensors")
        bufs = (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None for t in buffer])
        on_dev = 1
    else:
        keep = _host_bytes(buffer)
        bufs = (C.c_void_p * n)(*([C.c_void_p(keep.ctypes.data)] + [None] * (n - 1)))
        n_bytes, on_dev = keep.size, 0
    if out is not None and _is_torch(out):
        import torch
        t_dt = torch.float32 if np_dt is np.float32 else torch.float64
        if not out.is_cuda or out.dtype != t_dt or out.numel() < n_lines * nfft or not out.is_contiguous():
            raise ValueError("out tensor has the wrong dtype/size")
        res, out_ptr, out_dev = out, out.data_ptr(), 1
    else:
        res = np.empty((int(n_lines), int(nfft)), dtype=np_dt) if out is None else out
        if (not isinstance(res, np.ndarray) or res.dtype != np_dt or res.size < n_lines * nfft or not res.flags.c_contiguous):
            raise ValueError("out array has the wrong dtype/size")
        out_ptr, out_dev = res.ctypes.data, 0
    st = lib.spec_waterfall_multi(ctxs, n, bufs, on_dev, int(n_bytes), int(start_byte), dt, int(nfft), hop, int(n_lines),
                                  window, out_fmt, float(eof_fill), out_ptr, out_dev, int(n_chunks))
    del keep
    services[0]._check(st)
    return res


def welch_psd_multi(services, buffer, start_byte: int, datatype: str, fs: float, nfft: int, hop: int, n_seg: int,
                    n_psd: int, psd_stride_bytes: int, window: int = L.WIN_HANN, scaling: int = L.PSD_DENSITY,
                    db: bool = False, out=None):
    """``spec_welch_psd_multi``: a batch of PSDs (``SpectralService.welch_psd`` with ``n_psd`` > 1) spread over several
    contexts -- the PSDs are independent, context r takes ``shard_lines(n_psd, len(services), r)`` of them, one host
    thread per context inside the library, nothing exchanged on the data path.

    ``buffer``: host bytes (every context stages the span of its own PSDs), or a LIST of CUDA uint8 tensors, entry r
    on ``services[r]``'s device starting at the first byte of shard r's first PSD (None for a shard without PSDs).
    ``out``: None / numpy float32 array -> host result; a CUDA float32 tensor on ``services[0]``'s device -> the peers
    send their rows there.  Returns ``(freq[nfft], psd[n_psd, nfft])`` -- what one context returns (bit for bit while both
    take the same form of the kernel; include/specgpu.h)."""
    lib = L.load()
    dt = dtype_from_sigmf(datatype)
    n = len(services)
    ctxs = (C.c_void_p * n)(*[s._ctx for s in services])
    keep = None
    if isinstance(buffer, (list, tuple)):
        import torch
        if len(buffer) != n:
            raise ValueError("device input: one tensor (or None) per service")
        for t in buffer:
            if t is not None and (not t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous()):
                raise ValueError("device buffers must be contiguous CUDA uint8 tensors")
        bufs = (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None for t in buffer])
        sizes = (C.c_uint64 * n)(*[int(t.numel()) if t is not None else 0 for t in buffer])
        on_dev = 1
    else:
        keep = _host_bytes(buffer)
        bufs = (C.c_void_p * n)(*([C.c_void_p(keep.ctypes.data)] + [None] * (n - 1)))
        sizes = (C.c_uint64 * n)(*([int(keep.size)] + [0] * (n - 1)))
        on_dev = 0
    if out is not None and _is_torch(out):
        import torch
        if not out.is_cuda or out.dtype != torch.float32 or out.numel() < n_psd * nfft or not out.is_contiguous():
            raise ValueError("out tensor has the wrong dtype/size")
        res, out_ptr, out_dev = out, out.data_ptr(), 1
    else:
        res = np.empty((int(n_psd), int(nfft)), dtype=np.float32) if out is None else out
        if not isinstance(res, np.ndarray) or res.dtype != np.float32 or res.size < n_psd * nfft or not res.flags.c_contiguous:
            raise ValueError("out array has the wrong dtype/size")
        out_ptr, out_dev = res.ctypes.data, 0
    freq = np.empty(max(int(nfft), 0), dtype=np.float64)
    st = lib.spec_welch_psd_multi(ctxs, n, bufs, on_dev, sizes, int(start_byte), int(psd_stride_bytes), int(n_psd), dt,
                                  int(nfft) & 0xFFFFFFFF, int(hop), int(n_seg), window, scaling, float(fs), int(db),
                                  freq.ctypes.data, out_ptr, out_dev)
    del keep
    services[0]._check(st)
    return freq, res


class SpectralService:
    """GPU-backed drop-in for the reference ``SpectralService`` singleton."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, ref_cf64_zero: bool = False,
                 ref_edc_cf64_stride8: bool = False):
        """``stream`` is a hipStream_t handle (e.g. ``torch.cuda.Stream().cuda_stream``);
        None means the device's default stream, so device-resident results are
        ordered with PyTorch work on its default stream.  Device-pointer calls
        are asynchronous on that stream."""
        self._lib = L.load()
        self._ctx = C.c_void_p()
        flags = L.FLAG_REF_CF64_ZERO if ref_cf64_zero else 0
        if ref_edc_cf64_stride8:
            flags |= L.FLAG_REF_EDC_CF64_STRIDE8
        if not stream:
            flags |= L.FLAG_NULL_STREAM
        st = self._lib.spec_create(int(device), C.c_void_p(stream) if stream else None, flags,
                                   C.byref(self._ctx))
        if st != L.SPEC_OK:
            msg = self._lib.spec_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise _ERRORS.get(st, RuntimeError)(msg)
        self.device = int(device)

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.spec_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st: int) -> None:
        if st != L.SPEC_OK:
            raise _ERRORS.get(st, RuntimeError)(self._lib.spec_last_error(self._ctx).decode())

    def synchronize(self) -> None:
        self._check(self._lib.spec_sync(self._ctx))

    def set_option(self, key: str, value: int) -> None:
        """Tuning / testing knobs of ``spec_set_option`` (include/specgpu.h)."""
        self._check(self._lib.spec_set_option(self._ctx, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        """``spec_get_option``: a knob's current value, or the read-only state ``large_team_disabled``."""
        v = C.c_int64(0)
        self._check(self._lib.spec_get_option(self._ctx, key.encode(), C.byref(v)))
        return int(v.value)

    @property
    def stream(self) -> int:
        return int(self._lib.spec_stream(self._ctx) or 0)

    # -- SpectralService.computeMagnitudes (SS:33-85) -----------------------
    def compute_magnitudes(self, buffer, start_byte: int, nfft: int, datatype: str,
                           big_endian: Optional[bool] = None) -> np.ndarray:
        """One spectrogram line as ``double[nfft]`` (index 0 = -fs/2).

        ``buffer`` stands for the ``MappedByteBuffer``; its byte order defaults
        to what ``SigMfHelper.load`` would set for ``datatype``
        (sigmf/SigMfHelper.java:87-91: "_le" suffix -> little endian, else big).
        Raises ValueError for a non power-of-two nfft (commons-math3's
        MathIllegalArgumentException) and IndexError when the slice leaves the
        buffer (the ByteBuffer getters' IndexOutOfBoundsException).
        """
        b = _host_bytes(buffer)
        if big_endian is None:
            big_endian = not datatype.endswith("_le")
        out = np.empty(int(nfft) if nfft > 0 else 0, dtype=np.float64)
        self._check(self._lib.spec_compute_magnitudes(
            self._ctx, b.ctypes.data, b.size, int(start_byte), int(nfft) & 0xFFFFFFFF,
            datatype.encode(), int(bool(big_endian)), out.ctypes.data))
        return out

    # -- MainController.updateDisplay slice loop (MC:980-999), batched ------
    def compute_waterfall(self, buffer, start_byte: int, nfft: int, datatype: str,
                          n_lines: int, hop: Optional[int] = None, window: int = L.WIN_RECT,
                          out_fmt: int = L.OUT_DB20_F32, eof_fill: float = -150.0, out=None):
        """``n_lines`` lines, ``hop`` samples apart (reference: hop = nfft).

        ``buffer`` is host bytes (numpy / bytes / mmap) or a torch uint8 CUDA
        tensor; the result has the same residency (numpy array, or a torch
        tensor on the context's device).  Lines past the end of the buffer are
        ``eof_fill`` (MC:994-998).
        """
        hop = int(nfft if hop is None else hop)
        dt = dtype_from_sigmf(datatype)
        np_dt = _OUT_NP[out_fmt]
        if _is_torch(buffer):
            import torch
            if not buffer.is_cuda or buffer.dtype != torch.uint8 or not buffer.is_contiguous():
                raise ValueError("device buffer must be a contiguous CUDA uint8 tensor")
            t_dt = torch.float32 if np_dt is np.float32 else torch.float64
            if out is None:
                out = torch.empty((int(n_lines), int(nfft)), dtype=t_dt, device=buffer.device)
            elif out.dtype != t_dt or out.numel() < n_lines * nfft or not out.is_contiguous():
                raise ValueError("out tensor has the wrong dtype/size")
            self._check(self._lib.spec_waterfall(
                self._ctx, buffer.data_ptr(), 1, buffer.numel(), int(start_byte), dt, int(nfft), hop,
                int(n_lines), window, out_fmt, float(eof_fill), out.data_ptr(), 1))
            return out
        b = _host_bytes(buffer)
        if out is None:
            res = np.empty((int(n_lines), int(nfft)), dtype=np_dt)
        else:
            res = out
            if (not isinstance(res, np.ndarray) or res.dtype != np_dt or res.size < n_lines * nfft
                    or not res.flags.c_contiguous):
                raise ValueError("out array has the wrong dtype/size")
        self._check(self._lib.spec_waterfall(
            self._ctx, b.ctypes.data, 0, b.size, int(start_byte), dt, int(nfft), hop, int(n_lines),
            window, out_fmt, float(eof_fill), res.ctypes.data, 0))
        return res

    # -- MainController.renderSpectrogram (MC:1261-1291) + getColorForMagnitude (MC:926-957) ----
    def render_spectrogram(self, tile, height: int, fs: float, min_db: float = -100.0, max_db: float = 0.0,
                           colormap: int = L.CMAP_GRAYSCALE):
        """dB tile ``[width, nfft]`` (numpy float32 or CUDA tensor) -> BGRA8 image ``[height, width, 4]``."""
        if _is_torch(tile):
            import torch
            if not tile.is_cuda or tile.dtype != torch.float32 or not tile.is_contiguous():
                raise ValueError("tile must be a contiguous CUDA float32 tensor")
            out = torch.empty((int(height), tile.shape[0], 4), dtype=torch.uint8, device=tile.device)
            self._check(self._lib.spec_render_spectrogram(
                self._ctx, tile.data_ptr(), 1, tile.shape[0], tile.shape[1], int(height), float(fs), float(min_db),
                float(max_db), colormap, out.data_ptr(), 1))
            return out
        t = np.ascontiguousarray(tile, dtype=np.float32)
        out = np.empty((int(height), t.shape[0], 4), dtype=np.uint8)
        self._check(self._lib.spec_render_spectrogram(
            self._ctx, t.ctypes.data, 0, t.shape[0], t.shape[1], int(height), float(fs), float(min_db),
            float(max_db), colormap, out.ctypes.data, 0))
        return out

    def waterfall_render(self, buffer, start_byte: int, nfft: int, datatype: str, width: int, height: int,
                         fs: float, hop: Optional[int] = None, window: int = L.WIN_RECT, min_db: float = -100.0,
                         max_db: float = 0.0, colormap: int = L.CMAP_GRAYSCALE) -> np.ndarray:
        """One redraw of ``MainController.updateDisplay()`` (MC:962-1049): ``width`` lines from
        ``start_byte`` rendered to a BGRA8 image; the dB tile stays on the device."""
        hop = int(nfft if hop is None else hop)
        if _is_torch(buffer):  # device-resident recording -> device-resident image
            import torch
            if not buffer.is_cuda or buffer.dtype != torch.uint8 or not buffer.is_contiguous():
                raise ValueError("device buffer must be a contiguous CUDA uint8 tensor")
            img = torch.empty((int(height), int(width), 4), dtype=torch.uint8, device=buffer.device)
            self._check(self._lib.spec_waterfall_render(
                self._ctx, buffer.data_ptr(), 1, buffer.numel(), int(start_byte), dtype_from_sigmf(datatype), int(nfft),
                hop, int(width), window, int(height), float(fs), float(min_db), float(max_db), colormap,
                img.data_ptr(), 1))
            return img
        b = _host_bytes(buffer)
        out = np.empty((int(height), int(width), 4), dtype=np.uint8)
        self._check(self._lib.spec_waterfall_render(
            self._ctx, b.ctypes.data, 0, b.size, int(start_byte), dtype_from_sigmf(datatype), int(nfft), hop,
            int(width), window, int(height), float(fs), float(min_db), float(max_db), colormap,
            out.ctypes.data, 0))
        return out

    def count_lines(self, n_bytes: int, start_byte: int, datatype: str, nfft: int, hop: int) -> int:
        return int(self._lib.spec_count_lines(int(n_bytes), int(start_byte), dtype_from_sigmf(datatype),
                                              int(nfft), int(hop)))

    # -- PowerSpectralDensity.calculatePsdWelch call site (ADC:303-313) -----
    def welch_psd(self, buffer, start_byte: int, datatype: str, fs: float, nfft: int = 8192,
                  hop: Optional[int] = None, n_seg: Optional[int] = None, window: int = L.WIN_HANN,
                  scaling: int = L.PSD_DENSITY, db: bool = False, n_psd: int = 1,
                  psd_stride_bytes: int = 0, out=None) -> Tuple[np.ndarray, object]:
        """Welch PSD; returns ``(freq[nfft], psd[n_psd, nfft])`` like the two
        rows the reference plots (ADC:324-328).  ``hop`` defaults to nfft/2 and
        ``n_seg`` to every whole segment available after ``start_byte``."""
        hop = int(max(nfft // 2, 1) if hop is None else hop)
        dt = dtype_from_sigmf(datatype)
        on_dev = _is_torch(buffer)
        n_bytes = buffer.numel() if on_dev else None
        if on_dev:
            import torch
            if not buffer.is_cuda or buffer.dtype != torch.uint8 or not buffer.is_contiguous():
                raise ValueError("device buffer must be a contiguous CUDA uint8 tensor")
            ptr = buffer.data_ptr()
        else:
            b = _host_bytes(buffer)
            n_bytes, ptr = b.size, b.ctypes.data
        if n_seg is None:
            n_seg = int(self._lib.spec_count_lines(n_bytes, int(start_byte), dt, int(nfft), hop))
        freq = np.empty(int(nfft), dtype=np.float64)
        if on_dev:
            if out is None:
                psd = torch.empty((int(n_psd), int(nfft)), dtype=torch.float32, device=buffer.device)
            else:
                psd = out
                if (not _is_torch(psd) or not psd.is_cuda or psd.dtype != torch.float32 or not psd.is_contiguous()
                        or psd.numel() < n_psd * nfft):
                    raise ValueError("out tensor has the wrong dtype/size")
            out_ptr = psd.data_ptr()
        else:
            if out is not None:
                raise ValueError("out= is for device-resident buffers; with host bytes the PSDs come back as a numpy array")
            psd = np.empty((int(n_psd), int(nfft)), dtype=np.float32)
            out_ptr = psd.ctypes.data
        self._check(self._lib.spec_welch_psd(
            self._ctx, ptr, int(on_dev), n_bytes, int(start_byte), int(psd_stride_bytes), int(n_psd), dt,
            int(nfft), hop, int(n_seg), window, scaling, float(fs), int(db), freq.ctypes.data, out_ptr,
            int(on_dev)))
        return freq, psd

    def calculate_psd_welch(self, data, fs: float, nfft: int, hop: Optional[int] = None,
                            window: int = L.WIN_HANN, scaling: int = L.PSD_DENSITY, db: bool = True):
        """The call at ADC:308-312, ``PowerSpectralDensity.calculatePsdWelch(data, fs, nfft)``:
        ``data`` is ``double[2][N]`` (row 0 = I, row 1 = Q); returns ``[freq, psd]`` like the
        reference's two rows (doubles, fp64 pipeline).  ``nfft`` is any integer >= 1: the dialog
        passes the burst length for bursts shorter than 8192 samples (ADC:303-307).  Window /
        overlap / scaling are explicit because JDSP's are not known (defaults: Hann, 50 %,
        density).  Row 1 is in dB (``10 log10(P + 1e-20)``) by default, because that is how the
        caller reads it: an additive dB offset (ADC:319-328), ``"%.1f dB"`` marker labels
        (ADC:612, 626), SNR = difference of two levels (ADC:675, 757), "dB/Hz" in the report
        (ADC:751); ``db=False`` returns the linear density."""
        keep, pre, pim, n, dev = self._planar(data)  # host double[2][N] or a device tensor (the down-converter's)
        hop = int(max(nfft // 2, 1) if hop is None else hop)
        freq = np.empty(max(int(nfft), 0), dtype=np.float64)
        psd = np.empty(max(int(nfft), 0), dtype=np.float64)
        self._check(self._lib.spec_welch_psd_planar_f64(
            self._ctx, pre, pim, dev, n, int(nfft) & 0xFFFFFFFF, hop, window,
            scaling, float(fs), int(db), freq.ctypes.data, psd.ctypes.data))
        del keep
        return np.stack([freq, psd])

    # -- Analysis dialog traces (ADC:219-284) --------------------------------
    def _planar(self, data):
        """``double[2][N]`` (row 0 = I, row 1 = Q) as two device pointers or two host arrays."""
        if _is_torch(data):
            import torch
            if data.dtype != torch.float64 or data.dim() != 2 or data.shape[0] != 2 or not data.is_contiguous():
                raise ValueError("expected a contiguous float64 tensor of shape [2, N]")
            n = int(data.shape[1])
            return data, data.data_ptr(), data.data_ptr() + 8 * n, n, 1
        a = np.ascontiguousarray(np.asarray(data, dtype=np.float64))
        if a.ndim != 2 or a.shape[0] != 2:
            raise ValueError("expected double[2][N]")
        return a, a[0].ctypes.data, a[1].ctypes.data, int(a.shape[1]), 0

    def _trace_out(self, like, n: int, on_device: int):
        if on_device:
            import torch
            out = torch.empty(n, dtype=torch.float64, device=like.device)
            return out, out.data_ptr()
        out = np.empty(n, dtype=np.float64)
        return out, out.ctypes.data

    def magnitude_trace(self, data, alpha: float):
        """``updateMagnitudeChart`` (ADC:219-246): ``20 log10`` of the EMA of ``hypot(I, Q)``,
        one value per sample (non-finite values are the caller's to drop, ADC:239-242)."""
        keep, pre, pim, n, dev = self._planar(data)
        out, pout = self._trace_out(keep, n, dev)
        self._check(self._lib.spec_magnitude_trace(self._ctx, pre, pim, dev, n, float(alpha), pout, dev))
        return out

    def inst_freq_trace(self, data, alpha: float, fs: float, center_freq: float = 0.0):
        """``updateFrequencyChart`` (ADC:256-284): EMA of the wrapped phase step in Hz plus the
        centre frequency, ``N - 1`` values (samples 1 .. N-1)."""
        keep, pre, pim, n, dev = self._planar(data)
        out, pout = self._trace_out(keep, max(n - 1, 0), dev)
        self._check(self._lib.spec_inst_freq_trace(self._ctx, pre, pim, dev, n, float(alpha), float(fs),
                                                   float(center_freq), pout, dev))
        return out

    # -- synthetic recording (bench / tests) --------------------------------
    def synth_iq(self, datatype: str, seed: int, first_sample: int, n_samples: int, out=None):
        """Counter-based synthetic IQ generated on the device (uint8 tensor)."""
        import torch
        nbytes = int(n_samples) * bytes_per_sample(datatype)
        if out is None:
            out = torch.empty(nbytes, dtype=torch.uint8, device="cuda:%d" % self.device)
        self._check(self._lib.spec_synth_iq(self._ctx, out.data_ptr(), dtype_from_sigmf(datatype),
                                            int(seed), int(first_sample), int(n_samples)))
        return out
