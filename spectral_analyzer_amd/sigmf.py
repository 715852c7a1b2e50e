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
