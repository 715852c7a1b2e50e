"""Host-side mirror of the reference's SigMF loader, as far as the hot path needs it.

Reference: ``sigmf/SigMfHelper.java:43-94`` (load), ``sigmf/Global.java:19-79``
(``core:datatype``, ``core:sample_rate`` default 1e6, ``core:dataset``,
``getBytesPerSample``), ``sigmf/Capture.java:17-43`` (``core:header_bytes`` of the
first capture) and ``controllers/MainController.java:603-605`` (total samples).

Kept: which file holds the samples (``core:dataset`` relative to the meta file, else
the ``.sigmf-meta`` -> ``.sigmf-data`` rename), the header skip, the byte order rule
("_le" suffix -> little endian, anything else big endian), the datatype string handed
to ``computeMagnitudes``.  Lifted: the 2 GiB mapping cap (``SigMfHelper.java:78-82``,
``int`` offsets at ``MainController.java:985``) -- the map covers the whole file and
all offsets are 64-bit.  Annotations, JSON round-tripping and the rest of the records
stay in the Java application.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib as L
from .spectral_service import SpectralService, bytes_per_sample


@dataclass
class SigMfRecording:
    meta_path: str
    data_path: str
    datatype: str          # Global.datatype(), e.g. "ci16_le"
    sample_rate: float     # Global.sampleRate(), default 1e6 (Global.java:40-42)
    header_bytes: int      # first capture's core:header_bytes (SigMfHelper.java:60-67)
    buffer: np.ndarray     # uint8 view of the data file after the header (the MappedByteBuffer)

    @property
    def big_endian(self) -> bool:
        """SigMfHelper.java:87-91: "_le" suffix -> LITTLE_ENDIAN, else BIG_ENDIAN."""
        return not self.datatype.endswith("_le")

    @property
    def bytes_per_sample(self) -> int:
        return bytes_per_sample(self.datatype)

    @property
    def total_samples(self) -> int:
        """MainController.java:603-605."""
        return int(self.buffer.size) // self.bytes_per_sample

    # -- MainController.updateDisplay (MC:962-999) on this recording ------------------
    def waterfall(self, svc: SpectralService, current_sample_offset: int, fft_size: int, canvas_w: int,
                  hop: Optional[int] = None, window: int = L.WIN_RECT, out_fmt: int = L.OUT_DB20_F32):
        """``canvas_w`` lines starting at ``current_sample_offset`` (MC:984: line t starts at
        ``offset + t * fftSize``; ``hop`` generalises fftSize); lines past the end are -150.0."""
        start_byte = int(current_sample_offset) * self.bytes_per_sample          # MC:985, 64-bit here
        return svc.compute_waterfall(self.buffer, start_byte, fft_size, self.datatype, canvas_w, hop=hop,
                                     window=window, out_fmt=out_fmt)

    def compute_magnitudes(self, svc: SpectralService, byte_offset: int, fft_size: int) -> np.ndarray:
        """The single call of MC:988-993."""
        return svc.compute_magnitudes(self.buffer, byte_offset, fft_size, self.datatype, self.big_endian)

    # -- the same through the library's own file reader (include/specgpu.h, spec_open_recording) --------
    def open_native(self, svc: SpectralService) -> "NativeRecording":
        """The data file handed to the library by PATH (``spec_open_recording``): the library maps the whole
        file itself (64-bit length; option ``rec_pread`` = 1: preads the slices into a pinned ring instead) --
        what a Java host uses instead of the <= 2 GiB ``MappedByteBuffer`` of ``SigMfHelper.java:78-84``."""
        return NativeRecording(svc, self)


class NativeRecording:
    """``spec_recording`` handle of one data file; close it (or use ``with``) before the service."""

    def __init__(self, svc: SpectralService, rec: SigMfRecording):
        import ctypes as C
        self._svc, self.rec = svc, rec
        self._h = C.c_void_p()
        svc._check(svc._lib.spec_open_recording(svc._ctx, os.fsencode(rec.data_path), int(rec.header_bytes),
                                                C.byref(self._h)))

    @property
    def n_bytes(self) -> int:
        return int(self._svc._lib.spec_recording_bytes(self._h))

    def waterfall(self, current_sample_offset: int, fft_size: int, canvas_w: int, hop: Optional[int] = None,
                  window: int = L.WIN_RECT, out_fmt: int = L.OUT_DB20_F32, eof_fill: float = -150.0) -> np.ndarray:
        """MC:980-999 on the file itself: line t starts at sample ``offset + t * hop``."""
        hop = int(fft_size if hop is None else hop)
        out = np.empty((int(canvas_w), int(fft_size)), dtype=np.float64 if out_fmt >= L.OUT_DB20_F64 else np.float32)
        svc = self._svc
        svc._check(svc._lib.spec_waterfall_recording(
            svc._ctx, self._h, int(current_sample_offset) * self.rec.bytes_per_sample,
            svc._lib.spec_dtype_from_sigmf(self.rec.datatype.encode()), int(fft_size), hop, int(canvas_w), window,
            out_fmt, float(eof_fill), out.ctypes.data, 0))
        return out

    def compute_magnitudes(self, byte_offset: int, fft_size: int) -> np.ndarray:
        """SS:33-85 with a 64-bit ``startByte``."""
        out = np.empty(max(int(fft_size), 0), dtype=np.float64)
        svc = self._svc
        svc._check(svc._lib.spec_compute_magnitudes_recording(
            svc._ctx, self._h, int(byte_offset), int(fft_size) & 0xFFFFFFFF, self.rec.datatype.encode(),
            int(self.rec.big_endian), out.ctypes.data))
        return out

    def close(self) -> None:
        if self._h is not None and self._h.value:
            self._svc._lib.spec_close_recording(self._h)
            self._h = None

    def __del__(self):  # an unclosed handle would keep the descriptor and the whole-file mapping
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def load(meta_path: str) -> SigMfRecording:
    """``SigMfHelper.load`` (SigMfHelper.java:43-94) without the 2 GiB cap."""
    meta_path = os.fspath(meta_path)
    with open(meta_path, "r", encoding="utf-8") as f:
        meta = json.load(f)                                                       # SMH:45 (unknown keys ignored)
    g = meta.get("global") or {}
    datatype = g.get("core:datatype")
    if datatype is None:
        raise ValueError("core:datatype missing in " + meta_path)
    sample_rate = g.get("core:sample_rate")
    sample_rate = 1000000.0 if sample_rate is None else float(sample_rate)        # Global.java:40-42
    parent = os.path.dirname(meta_path)
    if g.get("core:dataset") is not None:                                         # SMH:49-53
        data_path = os.path.join(parent, g["core:dataset"])
    else:                                                                         # SMH:54-57
        data_path = meta_path.replace(".sigmf-meta", ".sigmf-data")
    header = 0
    caps = meta.get("captures") or []
    if caps and caps[0].get("core:header_bytes") is not None:                     # SMH:60-67
        header = int(caps[0]["core:header_bytes"])
    size = os.path.getsize(data_path)
    avail = max(0, size - header)                                                 # SMH:76
    if avail == 0:
        buf = np.zeros(0, dtype=np.uint8)
    else:
        buf = np.memmap(data_path, dtype=np.uint8, mode="r", offset=header, shape=(avail,))   # SMH:84, no 2 GiB cap
    return SigMfRecording(meta_path, data_path, datatype, sample_rate, header, buf)
