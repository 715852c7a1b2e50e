"""Host-side mirror of the reference's ``ExtractDownConvertService`` for the GPU path.

Reference: ``services/ExtractDownConvertService.java:54-117`` --
``double[][] extractAndDownConvert(MappedByteBuffer buffer, long startSample, int count,
String datatype, double freqOff, int down, boolean fast)``: the burst reader, the frequency
shift and the decimating filter behind the Analysis dialog.  The reader is the reference's
own arithmetic (bit-exact); the filter is JDSP's ``Resampler`` (not in the reference tree), so
the library implements its own stated specification (``include/specgpu.h``,
``spec_down_convert``) -- parity with JDSP is unpinned.
"""
from __future__ import annotations

import numpy as np

from . import _lib as L
from .spectral_service import SpectralService, _host_bytes, _is_torch, dtype_from_sigmf


class ExtractDownConvertService:
    """Shares the device context of a :class:`SpectralService` (the two reference services are
    singletons of the same application)."""

    def __init__(self, service: SpectralService):
        self._svc = service

    def _buffer(self, buffer):
        if _is_torch(buffer):
            return buffer, buffer.data_ptr(), buffer.numel() * buffer.element_size(), 1
        b = _host_bytes(buffer)
        return b, b.ctypes.data, b.size, 0

    def _out(self, like, n: int, dev: int):
        if dev:
            import torch
            out = torch.empty((2, n), dtype=torch.float64, device=like.device)
            return out, out.data_ptr(), out.data_ptr() + 8 * n
        out = np.empty((2, n), dtype=np.float64)
        return out, out[0].ctypes.data, out[1].ctypes.data

    def extract_iq(self, buffer, start_sample: int, count: int, datatype: str):
        """The reader alone (EDC:60-97): ``double[2][count]``, row 0 = I, row 1 = Q.
        Host buffer -> numpy result, device tensor -> device result."""
        if count < 0 or start_sample < 0:
            raise IndexError("negative sample range")
        keep, ptr, cap, dev = self._buffer(buffer)
        out, pre, pim = self._out(keep, int(count), dev)
        s = self._svc
        s._check(s._lib.spec_extract_iq(s._ctx, ptr, dev, cap, int(start_sample), int(count),
                                        dtype_from_sigmf(datatype), pre, pim, dev))
        return out

    def extract_and_down_convert(self, buffer, start_sample: int, count: int, datatype: str,
                                 freq_off: float, down: int, fast: bool):
        """``extractAndDownConvert`` (EDC:54-117): ``double[2][count // down]``.
        ``freq_off`` is in cycles per input sample (the reference passes a rate of 1.0)."""
        if count < 0 or start_sample < 0:
            raise IndexError("negative sample range")
        if down <= 0:
            raise ValueError("down must be >= 1")
        keep, ptr, cap, dev = self._buffer(buffer)
        out, pre, pim = self._out(keep, int(count) // int(down), dev)
        s = self._svc
        s._check(s._lib.spec_down_convert(s._ctx, ptr, dev, cap, int(start_sample), int(count),
                                          dtype_from_sigmf(datatype), float(freq_off), int(down),
                                          L.DC_FAST if fast else L.DC_LPF, pre, pim, dev))
        return out
